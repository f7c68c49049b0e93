// Collapses the host's BVH2 into the 8-wide BVH with quantised child boxes that the device's single-level traversal
// kernels walk (include/tungsten_hip.h: TgHipWideNode).  Takes the place of Embree's BVH4 build for the reference
// (thirdparty/embree/kernels/bvh/bvh_builder_sah.cpp via rtcCommit, primitives/TriangleMesh.cpp:565).
#ifndef TGAMD_WIDEBVH_HPP_
#define TGAMD_WIDEBVH_HPP_

#include "Math.hpp"
#include "../../../include/tungsten_hip.h"

#include <vector>

namespace tungsten_amd {

struct WideBvhResult
{
    std::vector<TgHipWideNode> nodes;  // nodes[0] is the root; empty when the BVH2 cannot be collapsed (a leaf of more than TGHIP_WIDE_MAX_LEAF records)
    std::vector<uint32_t> order;       // order[i] = record (in the BVH2's order) placed at slot i of the wide tree's record order
    int depth = 0;                     // nodes on the longest root-to-leaf path
};

// bvh2: the tree over records [0, recBounds.size()), root = bvh2[0]; recBounds[i] = box of record i.  The records have to be
// permuted by `order` afterwards (leaf children of one wide node are contiguous, wide nodes breadth first); the leaf
// references of bvh2 are rewritten here for that order, so both trees describe the same permuted record array.
WideBvhResult buildWideBvh(std::vector<TgHipBvhNode> &bvh2, const std::vector<Box3f> &recBounds);

} // namespace tungsten_amd

#endif
