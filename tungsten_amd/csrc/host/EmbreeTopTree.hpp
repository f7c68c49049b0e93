// The reference's top-level acceleration structure, restated: TraceableScene commits ONE Embree user geometry whose items are the scene's
// finite primitives (renderer/TraceableScene.hpp:112-134), and Embree 2.11 builds a BVH4 with one item per leaf over their bounds() --
// BVH4VirtualSceneBuilderSAH (kernels/bvh/bvh_builder_sah.cpp:811-815: sahBlockSize 4, leaf size 1) through the binned-SAH builder
// (kernels/builders/bvh_builder_sah.h:176-282, heuristic_binning.h, heuristic_binning_array_aligned.h).  Where faces coincide the ORDER in
// which a ray visits that tree's leaves decides which primitive it hits (include/tungsten_hip.h: TgHipTopNode), so the tree is part of the
// path's arithmetic: this builder produces it node for node -- same children in the same slots with the same boxes -- from the items'
// boxes.  tests/test_top_tree.py holds it to trees read out of the reference's own Embree (tests/golden/top_trees.json).
#pragma once
#include <cstdint>
#include <vector>
#include "../../../include/tungsten_hip.h"

namespace tungsten_amd {

struct TopBox { float lo[3], hi[3]; };

// The BVH4 over items 0 .. boxes.size()-1 in preorder (node 0 = the root; a child >= 0 is a node, < 0 the item ~child, TGHIP_TOP_EMPTY an
// unused slot).  Empty for fewer than two items (Embree's root is then the leaf itself: no box is tested) or when a box is not valid
// by Embree's rule (lower <= upper, finite), in which case Embree drops the item and the caller must not use a tree.
std::vector<TgHipTopNode> buildEmbreeTopTree(const std::vector<TopBox> &boxes);

// Quad::bounds (Quad.cpp:281-289), Cube::bounds (Cube.cpp:333-344), Sphere::bounds (Sphere.cpp:273-276), Disk::bounds (Disk.cpp:298-306) and
// Cylinder::bounds (Cylinder.cpp:272-279) from the flattened object: the box
// the reference's bounds callback reports for the record's primitive.  False for a record kind whose bounds are not restated (triangles, instances).
bool referenceLeafBounds(const TgHipObject &o, uint32_t kind, float lo[3], float hi[3]);

// The tree for a flattened scene: a flat list (<= TGHIP_FLAT_MAX_RECS records, no instances) whose records are all quads, cubes, spheres, disks or cylinders,
// one per object.  The items are the objects that have a record, in object order -- the order of the reference's _finites --, a leaf names the
// item's record.  Empty when the scene is not such a list (or has a single record: Embree's root is then the leaf).
std::vector<TgHipTopNode> buildSceneTopTree(const TgHipObject *objects, uint32_t numObjects, const TgHipPrimRec *recs, uint32_t numRecs);

}
