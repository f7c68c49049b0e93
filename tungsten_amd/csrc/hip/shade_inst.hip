// Explicit instantiations of k_shade variants (see pt_wavefront.h); the extern "C" shim in tungsten_hip.hip launches them.
// The per-family class variants of scenes with instance records.
#include "pt_wavefront.h"

template __global__ void k_shade<MASK_COAT_INST, 2, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_COAT_INST | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_GLASS_INST, 2, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_GLASS_INST | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
