// Explicit instantiation of the one k_shade variant that reads the scene's small tables from global memory (see pt_wavefront.h: GLOBAL_TABLES;
// pt_kernels.h: stageSceneTables): what shades every class of a scene whose tables do not fit the workgroups' LDS copy.
#include "pt_wavefront.h"

template __global__ void k_shade<BSDF_MASK_ALL, 2, 0, true>(DeviceScene, PathState, PassParams, int);
