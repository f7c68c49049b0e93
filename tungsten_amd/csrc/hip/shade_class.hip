// Explicit instantiations of k_shade variants (see pt_wavefront.h); the extern "C" shim in tungsten_hip.hip launches them.
#include "pt_wavefront.h"

template __global__ void k_shade<MASK_COAT, COAT_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_COAT | FEAT_QMC), COAT_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_COAT, COAT_WAVES, 2>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_COAT | FEAT_QMC), COAT_WAVES, 2>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_GLASS, 2, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_GLASS | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_GLASS, 2, 2>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_GLASS | FEAT_QMC), 2, 2>(DeviceScene, PathState, PassParams, int);
