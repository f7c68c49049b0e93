// Host BVH2 builder for the GPU traversal kernel: binned SAH (32 bins, like the reference's own
// builder in src/core/bvh/BinnedSahSplitter.hpp:15), one flat tree over every finite primitive
// record in world space.  Replaces Embree's BVH4 build (thirdparty/embree/kernels/bvh/
// bvh_builder_sah.cpp via rtcCommit in TriangleMesh.cpp:565 / TraceableScene.hpp:133).
//
// Output layout = include/tungsten_hip.h: 64-byte nodes holding both child boxes, children
// stored depth-first (child0 subtree directly follows its parent), leaves = contiguous runs of
// records (the caller permutes its record arrays with `order`).
#ifndef TGAMD_BVHBUILDER_HPP_
#define TGAMD_BVHBUILDER_HPP_

#include "Math.hpp"
#include "../../../include/tungsten_hip.h"

#include <vector>

namespace tungsten_amd {

struct BvhBuildResult
{
    std::vector<TgHipBvhNode> nodes;   // nodes[0] is the root
    std::vector<uint32_t> order;       // order[i] = input primitive placed at output slot i
    int maxDepth = 0;
    int maxLeafSize = 0;
    double sahCost = 0.0;
};

BvhBuildResult buildBvh(const std::vector<Box3f> &primBounds, int maxLeafSize = 4);

} // namespace tungsten_amd

#endif
