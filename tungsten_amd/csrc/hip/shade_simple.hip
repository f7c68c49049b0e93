// Explicit instantiations of k_shade variants (see pt_wavefront.h); the extern "C" shim in tungsten_hip.hip launches them.
#include "pt_wavefront.h"

template __global__ void k_shade<MASK_LEAN, LEAN_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_LEAN | FEAT_QMC), LEAN_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_LEAN, LEAN_WAVES, 3>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_LEAN | FEAT_QMC), LEAN_WAVES, 3>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_LEAN, LEAN_WAVES, 7>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_LEAN | FEAT_QMC), LEAN_WAVES, 7>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_SIMPLE, SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_SIMPLE | FEAT_QMC), SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_SIMPLE_INST, SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_SIMPLE_INST | FEAT_QMC), SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_SIMPLE, SIMPLE_WAVES, 3>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_SIMPLE | FEAT_QMC), SIMPLE_WAVES, 3>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<MASK_SIMPLE, SIMPLE_WAVES, 7>(DeviceScene, PathState, PassParams, int);
template __global__ void k_shade<(MASK_SIMPLE | FEAT_QMC), SIMPLE_WAVES, 7>(DeviceScene, PathState, PassParams, int);
