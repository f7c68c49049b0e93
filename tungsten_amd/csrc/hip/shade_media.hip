// Explicit instantiation of the lean media shading variant (see pt_wavefront.h: MASK_MEDIA); the extern "C" shim in tungsten_hip.hip launches it.
#include "pt_wavefront.h"

template __global__ void k_shade<MASK_MEDIA, 2, 0>(DeviceScene, PathState, PassParams, int);
