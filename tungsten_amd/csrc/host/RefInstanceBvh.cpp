// PROVENANCE.  This file is a RESTATEMENT of reference code, not an independent design: it rebuilds, node for node, the binary BVH that
// Tungsten's `Instance` primitive builds over its instances (src/core/bvh/BvhBuilder.cpp, FullSahSplitter.hpp, BinnedSahSplitter.hpp,
// BinaryBvh.hpp), because `Instance::intersect` answers "the LAST hit in THAT tree's visiting order" (src/core/primitives/Instance.cpp:290-311):
// the tree is part of the arithmetic of the path, and bit-parity with the reference needs the same tree.  Function and variable names follow the
// reference so that the two can be read side by side.  It is an altered version of that code, plainly marked as such, under its licence:
//
//   Tungsten -- Copyright (c) 2014 Benedikt Bitterli <benedikt.bitterli (at) gmail (dot) com>
//   This software is provided 'as-is', without any express or implied warranty.  In no event will the authors be held liable for any damages
//   arising from the use of this software.  Permission is granted to anyone to use this software for any purpose, including commercial
//   applications, and to alter it and redistribute it freely, subject to the following restrictions:
//     1. The origin of this software must not be misrepresented; you must not claim that you wrote the original software.  If you use this
//        software in a product, an acknowledgment in the product documentation would be appreciated but is not required.
//     2. Altered source versions must be plainly marked as such, and must not be misrepresented as being the original software.
//     3. This notice may not be removed or altered from any source distribution.
//
// The pattern is confined to this file and EmbreeTopTree.cpp (set-up code on the host, nothing on the hot path) and is not meant to grow.
#include "RefInstanceBvh.hpp"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <memory>
#include <stdexcept>

namespace tungsten_amd {

namespace {

// math/Box.hpp: an empty box is (max, lowest); diagonal() clamps at zero, so an empty box has area 0
struct RBox
{
    float lo[3], hi[3];
    RBox() { for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }
    explicit RBox(const Box3f &b) { for (int k = 0; k < 3; ++k) { lo[k] = b.lo[k]; hi[k] = b.hi[k]; } }
    void grow(const RBox &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void growPoint(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void diagonal(float *d) const { for (int k = 0; k < 3; ++k) d[k] = std::max(hi[k] - lo[k], 0.0f); }
    float area() const { float d[3]; diagonal(d); return (d[0]*d[1] + d[1]*d[2] + d[2]*d[0])*2.0f; }   // Box.hpp:67-75
    int maxDim() const { float d[3]; diagonal(d); int idx = 0; float m = d[0]; for (int i = 1; i < 3; ++i) if (d[i] > m) { m = d[i]; idx = i; } return idx; }   // Vec.hpp:369-380
};

struct Prim { RBox box; float centroid[3]; uint32_t id; float area; };   // bvh/Primitive.hpp

struct SplitInfo { RBox lBox, rBox, lCentroidBox, rCentroidBox; int dim; uint32_t idx; float cost; };   // bvh/Splitter.hpp

const float IntersectionCost = 1.0f, TraversalCost = 1.0f;

// ---- bvh/FullSahSplitter.hpp -------------------------------------------------------------------------------------------------------
struct FullSahSplitter
{
    static void sortPrims(uint32_t start, uint32_t end, int dim, std::vector<Prim> &prims)
    {
        std::sort(prims.begin() + start, prims.begin() + end + 1, [dim](const Prim &a, const Prim &b) {
            if (a.centroid[dim] == b.centroid[dim])
                return a.id < b.id;
            return a.centroid[dim] < b.centroid[dim];
        });
    }
    static void computeAreas(uint32_t start, uint32_t end, std::vector<Prim> &prims)
    {
        RBox rBox;
        for (uint32_t i = end; i > start; --i) {
            rBox.grow(prims[i].box);
            prims[i].area = rBox.area();
        }
        rBox.grow(prims[start].box);
        prims[start].area = rBox.area();
    }
    static void findSahSplit(uint32_t start, uint32_t end, int dim, std::vector<Prim> &prims, SplitInfo &split)
    {
        sortPrims(start, end, dim, prims);
        computeAreas(start, end, prims);
        RBox lBox(prims[start].box);
        for (uint32_t i = start + 1; i <= end; ++i) {
            const float cost = IntersectionCost*(lBox.area()*float(i - start) + prims[i].area*float(end - i + 1));
            if (cost < split.cost) {
                split.dim = dim;
                split.idx = i;
                split.lBox = lBox;
                split.cost = cost;
            }
            lBox.grow(prims[i].box);
        }
        if (split.dim == dim) {
            RBox rBox;
            for (uint32_t i = split.idx; i <= end; ++i)
                rBox.grow(prims[i].box);
            split.rBox = rBox;
        }
    }
    static void twoWaySahSplit(uint32_t start, uint32_t end, std::vector<Prim> &prims, const RBox &geomBox, SplitInfo &split)
    {
        split.dim = -1;
        split.cost = geomBox.area()*(float(end - start + 1)*IntersectionCost - TraversalCost);
        findSahSplit(start, end, 0, prims, split);
        findSahSplit(start, end, 1, prims, split);
        findSahSplit(start, end, 2, prims, split);
        if (split.dim == -1) {               // SAH split failed: midpoint split along the largest extent
            split.dim = geomBox.maxDim();
            split.idx = (end - start + 1)/2 + start;
            sortPrims(start, end, split.dim, prims);
            for (uint32_t i = start; i <= end; ++i)
                (i < split.idx ? split.lBox : split.rBox).grow(prims[i].box);
        } else if (split.dim != 2) {
            sortPrims(start, end, split.dim, prims);
        }
    }
};

// ---- bvh/BinnedSahSplitter.hpp -----------------------------------------------------------------------------------------------------
struct BinnedSahSplitter
{
    enum { BinCount = 32 };
    RBox geomBounds[3][BinCount], centroidBounds[3][BinCount];
    float centroidMin[3], centroidSpan[3];
    int counts[3][BinCount];

    BinnedSahSplitter() { std::memset(counts, 0, sizeof(counts)); }

    int primitiveBin(const Prim &prim, int dim) const
    {
        const int b = int(float(BinCount)*((prim.centroid[dim] - centroidMin[dim])/centroidSpan[dim]));
        return std::min(std::max(b, 0), BinCount - 1);
    }
    void partialBin(uint32_t start, uint32_t end, std::vector<Prim> &prims, const RBox &centroidBox)
    {
        for (int k = 0; k < 3; ++k) centroidMin[k] = centroidBox.lo[k];
        centroidBox.diagonal(centroidSpan);
        for (int dim = 0; dim < 3; ++dim)
            if (centroidSpan[dim] > 0.0f)
                for (uint32_t i = start; i <= end; ++i) {
                    const int idx = primitiveBin(prims[i], dim);
                    geomBounds[dim][idx].grow(prims[i].box);
                    centroidBounds[dim][idx].growPoint(prims[i].centroid);
                    counts[dim][idx]++;
                }
    }
    void findSahSplit(int dim, SplitInfo &split)
    {
        int rCount = 0;
        int rCounts[BinCount];
        RBox rBox;
        RBox rBoxes[BinCount];
        for (int i = BinCount - 1; i > 0; --i) {
            rCount += counts[dim][i];
            rBox.grow(geomBounds[dim][i]);
            rCounts[i] = rCount;
            rBoxes[i] = rBox;
        }
        int lCount = counts[dim][0];
        RBox lBox = geomBounds[dim][0];
        for (int i = 1; i < BinCount; ++i) {
            const float cost = IntersectionCost*(lBox.area()*float(lCount) + rBoxes[i].area()*float(rCounts[i]));
            if (cost < split.cost) {
                split.dim = dim;
                split.idx = uint32_t(i);
                split.cost = cost;
            }
            lCount += counts[dim][i];
            lBox.grow(geomBounds[dim][i]);
        }
    }
    uint32_t sortByBin(uint32_t start, uint32_t end, std::vector<Prim> &prims, int dim, int bin)
    {
        uint32_t left = start, right = end;
        while (left < right) {
            while (left < right && primitiveBin(prims[left], dim) < bin)
                left++;
            while (right > left && primitiveBin(prims[right], dim) >= bin)
                right--;
            if (left != right)
                std::swap(prims[left], prims[right]);
        }
        if (left == end || right == start)       // degenerate case: one past the end
            return end + 1;
        return left;
    }
    void twoWaySahSplit(uint32_t start, uint32_t end, std::vector<Prim> &prims, const RBox &box, SplitInfo &split)
    {
        split.dim = -1;
        split.cost = box.area()*(float(end - start + 1)*IntersectionCost - TraversalCost);
        for (int i = 0; i < 3; ++i)
            if (centroidSpan[i] > 0.0f)
                findSahSplit(i, split);
        if (split.dim == -1) {               // SAH split failed: midpoint split along the largest extent
            split.dim = box.maxDim();
            split.idx = BinCount/2;
        }
        const int bin = int(split.idx);
        split.idx = sortByBin(start, end, prims, split.dim, bin);
        const bool spanIsZero = centroidSpan[0] == 0.0f && centroidSpan[1] == 0.0f && centroidSpan[2] == 0.0f;   // Vec == scalar: every component
        if (split.idx > end || spanIsZero) {
            split.idx = start + (end - start + 1)/2;
            split.lBox = prims[start].box;
            split.rBox = prims[end].box;
            split.lCentroidBox = RBox(); split.lCentroidBox.growPoint(prims[start].centroid);
            split.rCentroidBox = RBox(); split.rCentroidBox.growPoint(prims[end].centroid);
            for (uint32_t i = start + 1; i < end; ++i) {
                if (i < split.idx) {
                    split.lBox.grow(prims[i].box);
                    split.lCentroidBox.growPoint(prims[i].centroid);
                } else {
                    split.rBox.grow(prims[i].box);
                    split.rCentroidBox.growPoint(prims[i].centroid);
                }
            }
        } else {
            split.lBox = geomBounds[split.dim][0];
            split.rBox = geomBounds[split.dim][BinCount - 1];
            split.lCentroidBox = centroidBounds[split.dim][0];
            split.rCentroidBox = centroidBounds[split.dim][BinCount - 1];
            for (int i = 1; i < BinCount - 1; ++i) {
                if (i < bin) {
                    split.lBox.grow(geomBounds[split.dim][i]);
                    split.lCentroidBox.grow(centroidBounds[split.dim][i]);
                } else {
                    split.rBox.grow(geomBounds[split.dim][i]);
                    split.rCentroidBox.grow(centroidBounds[split.dim][i]);
                }
            }
        }
    }
};

// ---- bvh/BvhBuilder.cpp, bvh/NaiveBvhNode.hpp ---------------------------------------------------------------------------------------
struct NaiveNode
{
    std::unique_ptr<NaiveNode> child[2];
    RBox box;
    uint32_t id = 0;
    bool isLeaf() const { return !child[0]; }
};

void recursiveBuild(int &depth, NaiveNode &dst, uint32_t start, uint32_t end, std::vector<Prim> &prims, const RBox &geomBox, const RBox &centroidBox)
{
    depth = 1;
    dst.box = geomBox;
    const uint32_t numPrims = end - start + 1;
    if (numPrims == 1) {
        dst.id = prims[start].id;
    } else if (numPrims <= 2) {              // branchFactor 2: an internal node with its primitives as leaf children
        for (uint32_t i = start; i <= end; ++i) {
            dst.child[i - start].reset(new NaiveNode());
            dst.child[i - start]->box = prims[i].box;
            dst.child[i - start]->id = prims[i].id;
        }
    } else {
        // sahSplit with branchFactor 2 (BvhBuilder.cpp:62-98): one two-way split of the whole interval
        SplitInfo split;
        if (numPrims <= 64) {
            FullSahSplitter::twoWaySahSplit(start, end, prims, geomBox, split);
        } else {
            std::unique_ptr<BinnedSahSplitter> s(new BinnedSahSplitter());   // (50 KB of bins: off the stack)
            s->partialBin(start, end, prims, centroidBox);
            s->twoWaySahSplit(start, end, prims, geomBox, split);
        }
        const uint32_t starts[2] = {start, split.idx}, ends[2] = {split.idx - 1, end};
        const RBox geomBoxes[2] = {split.lBox, split.rBox}, centroidBoxes[2] = {split.lCentroidBox, split.rCentroidBox};
        for (int i = 0; i < 2; ++i) {
            dst.child[i].reset(new NaiveNode());
            int childDepth;
            recursiveBuild(childDepth, *dst.child[i], starts[i], ends[i], prims, geomBoxes[i], centroidBoxes[i]);
            depth = std::max(depth, childDepth + 1);
        }
    }
}

// ---- bvh/BinaryBvh.hpp:118-153 ------------------------------------------------------------------------------------------------------
void setJointBbox(TgHipInstNode &n, const RBox &l, const RBox &r)
{
    for (int k = 0; k < 3; ++k) {
        n.box[k*4 + 0] = l.lo[k]; n.box[k*4 + 1] = r.lo[k];
        n.box[k*4 + 2] = l.hi[k]; n.box[k*4 + 3] = r.hi[k];
    }
}
uint32_t flatten(std::vector<TgHipInstNode> &nodes, const NaiveNode *node, uint32_t head, uint32_t &tail, uint32_t &primIndex, std::vector<uint32_t> &primIndices)
{
    if (node->isLeaf()) {
        setJointBbox(nodes[head], node->box, node->box);
        nodes[head].left = primIndex; nodes[head].count = 1;
        primIndices[primIndex++] = node->id;
        return 1;
    }
    const uint32_t childIdx = tail;
    setJointBbox(nodes[head], node->child[0]->box, node->child[1]->box);
    nodes[head].left = childIdx; nodes[head].count = 0;
    tail += 2;
    const uint32_t lPrims = flatten(nodes, node->child[0].get(), childIdx + 0, tail, primIndex, primIndices);
    const uint32_t rPrims = flatten(nodes, node->child[1].get(), childIdx + 1, tail, primIndex, primIndices);
    if (lPrims + rPrims <= 2) {              // maxPrimsPerLeaf = 2: the pair becomes one leaf (its box stays the two children's)
        nodes[head].count = lPrims + rPrims;
        nodes[head].left = nodes[childIdx].left;
        tail = childIdx;
    }
    return lPrims + rPrims;
}

} // namespace

RefInstanceBvh buildRefInstanceBvh(const std::vector<Box3f> &boxes)
{
    RefInstanceBvh out;
    if (boxes.empty())
        return out;
    if (boxes.size() > (1u << 20))
        throw std::runtime_error("more than 2^20 instances in one `instances` primitive: the reference bins those in parallel tasks (BvhBuilder.cpp:38-58), which is not restated");
    std::vector<Prim> prims(boxes.size());
    RBox geomBounds, centroidBounds;
    for (size_t i = 0; i < boxes.size(); ++i) {
        prims[i].box = RBox(boxes[i]);
        for (int k = 0; k < 3; ++k)
            prims[i].centroid[k] = (boxes[i].lo[k] + boxes[i].hi[k])/2.0f;      // Box::center()
        prims[i].id = uint32_t(i);
        prims[i].area = prims[i].box.area();
        geomBounds.grow(prims[i].box);
        centroidBounds.growPoint(prims[i].centroid);
    }
    NaiveNode root;
    recursiveBuild(out.depth, root, 0, uint32_t(prims.size() - 1), prims, geomBounds, centroidBounds);
    // BinaryBvh::BinaryBvh: at most 2 n - 1 nodes before the pairs are merged
    out.nodes.resize(2*boxes.size());
    std::memset(out.nodes.data(), 0, out.nodes.size()*sizeof(TgHipInstNode));
    out.primIndices.resize(boxes.size());
    uint32_t tail = 1, primIndex = 0;
    flatten(out.nodes, &root, 0, tail, primIndex, out.primIndices);
    out.nodes.resize(tail);
    for (int k = 0; k < 3; ++k) { out.bounds.lo[k] = root.box.lo[k]; out.bounds.hi[k] = root.box.hi[k]; }
    return out;
}

} // namespace tungsten_amd
