// The BVH the reference's `instances` primitive builds over its instances and walks in Instance::intersect -- restated node for node.
//
// Instance::intersect (primitives/Instance.cpp:290-311) hands every instance whose box its BinaryBvh lets the ray into a ray with
// farT = infinity (Ray::scatter's default) and keeps the LAST hit in the tree's visiting order, so which surface a path sees depends on
// the order the reference's own tree is walked in -- its topology, its leaf pairs, its near/far rule.  A renderer that is to return the
// reference's result has to walk that very tree.  This file restates its construction:
//   Instance::prepareForRender      (Instance.cpp:392-428)       one bvh primitive per instance: box of the master box's eight rotated corners
//   Bvh::BvhBuilder(2).build        (bvh/BvhBuilder.cpp:29-201)  recursive two-way splits
//   Bvh::FullSahSplitter            (bvh/FullSahSplitter.hpp)    <= 64 primitives: exact SAH over three sorted orders
//   Bvh::BinnedSahSplitter          (bvh/BinnedSahSplitter.hpp)  <= 2^20 primitives: 32 bins per axis
//   Bvh::BinaryBvh(prims, 2)        (bvh/BinaryBvh.hpp:134-196)  flattened depth-first, sibling pairs adjacent, leaves of <= 2 primitives
// All arithmetic is float32 in the reference's order (areas, costs, bin indices), so splits and ties come out as they do there.
// The walk itself (BinaryBvh::trace, :197-287) is restated where it runs: oracle/oracle.c and csrc/hip/pt_kernels.h.
#ifndef TGAMD_REFINSTANCEBVH_HPP_
#define TGAMD_REFINSTANCEBVH_HPP_

#include "Math.hpp"
#include "../../../include/tungsten_hip.h"

#include <vector>

namespace tungsten_amd {

struct RefInstanceBvh
{
    std::vector<TgHipInstNode> nodes;      // nodes[0] is the root (a leaf when there are <= 2 instances)
    std::vector<uint32_t> primIndices;     // BinaryBvh::_primIndices: the instance behind leaf slot i
    Box3f bounds;                          // BinaryBvh::_bounds (the root's box)
    int depth = 0;                         // BvhBuilder::depth(): BinaryBvh::trace's stack holds depth + 1 entries
};

// boxes[i] = bGlobal of instance i (Instance.cpp:409-421); more than 2^20 instances are refused (the reference's parallel binning
// path, BvhBuilder.cpp:38-58, is not restated)
RefInstanceBvh buildRefInstanceBvh(const std::vector<Box3f> &boxes);

} // namespace tungsten_amd

#endif
