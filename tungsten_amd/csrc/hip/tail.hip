// Explicit instantiations of k_tail (see pt_wavefront.h); the extern "C" shim in tungsten_hip.hip launches them.
#include "pt_wavefront.h"

template __global__ void k_tail<MASK_TAIL, false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_tail<MASK_TAIL, true>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_tail<(MASK_TAIL | FEAT_QMC), false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_tail<(MASK_TAIL | FEAT_QMC), true>(DeviceScene, PathState, PassParams, uint32_t);
