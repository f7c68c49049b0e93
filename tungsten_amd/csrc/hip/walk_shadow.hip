// Explicit instantiations of the single-level shadow walk (pt_wavefront.h: k_trace_shadow_fast); the extern "C" shim in tungsten_hip.hip launches
// them.  A translation unit of its own so that it compiles next to the shim's (the longest one) under make -j; it was also where building
// without SLP vectorisation was found to pay (119 -> 98 VGPRs, 724 -> 634 us per launch: Makefile, profiles/r5_ab_no_slp.txt) before the
// whole library was built that way.
#include "pt_wavefront.h"

template __global__ void k_trace_shadow_fast<false, false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_trace_shadow_fast<false, true>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_trace_shadow_fast<true, false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_trace_shadow_fast<true, true>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_trace_shadow_fast_inst<false, false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_trace_shadow_fast_inst<false, true>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_trace_shadow_fast_inst<true, false>(DeviceScene, PathState, PassParams, uint32_t);
template __global__ void k_trace_shadow_fast_inst<true, true>(DeviceScene, PathState, PassParams, uint32_t);
