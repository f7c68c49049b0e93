// Explicit instantiations of k_tail (see pt_wavefront.h); the extern "C" shim in tungsten_hip.hip launches them.
#include "pt_wavefront.h"

template __global__ void k_tail<MASK_TAIL, false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_tail<MASK_TAIL, true>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_tail<(MASK_TAIL | FEAT_QMC), false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_tail<(MASK_TAIL | FEAT_QMC), true>(DeviceScene, PathState, PassParams, uint32_t);
// the tail of scenes whose materials are Lambert / null and the conductor family only (the metric's scene): a third of the instructions of the
// all-types variant, at one wave per SIMD where instruction latency is what a tail iteration takes
template __global__ void k_tail<MASK_COAT, false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_tail<(MASK_COAT | FEAT_QMC), false>(DeviceScene, PathState, PassParams, uint32_t);
