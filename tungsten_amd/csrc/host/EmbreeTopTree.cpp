// PROVENANCE.  This file is a scalar RE-DERIVATION of Embree 2.11's binned-SAH BVH4 builder for user geometry (as vendored by the reference:
// src/thirdparty/embree/kernels/builders/heuristic_binning.h, heuristic_binning_array_aligned.h, bvh_builder_sah.h, priminfo.h,
// kernels/common/accelset.h).  It exists because the ORDER in which a ray visits that tree decides which of two coincident faces the reference
// hits (include/tungsten_hip.h: TgHipTopNode): bit-parity needs the same tree, so the arithmetic and the tie-breaking of the original are
// followed step by step (in scalar float32 instead of SSE lanes).  Embree is Copyright 2009-2016 Intel Corporation, licensed under the Apache
// License, Version 2.0 (http://www.apache.org/licenses/LICENSE-2.0); this derived restatement is distributed under the same terms, "AS IS",
// WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND.  It is checked against trees read out of the reference's own Embree (tests/test_top_tree.py).
// Embree 2.11's BVH4 builder for user geometry, restated (see EmbreeTopTree.hpp for what and why; citations are paths under
// /root/reference/src/thirdparty/embree/).  Scalar float32 throughout, no contraction (Makefile: -ffp-contract=off).
#include "EmbreeTopTree.hpp"
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <limits>

namespace tungsten_amd {

namespace {

const float INF = std::numeric_limits<float>::infinity();

// embree::BBox3fa with its SSE min / max (minps / maxps return the SECOND operand unless the first compares smaller / greater)
struct Box {
    float lo[3], hi[3];
    void clear() { for (int k = 0; k < 3; ++k) { lo[k] = INF; hi[k] = -INF; } }
    void extend(const float *l, const float *h) { for (int k = 0; k < 3; ++k) { lo[k] = lo[k] < l[k] ? lo[k] : l[k]; hi[k] = hi[k] > h[k] ? hi[k] : h[k]; } }
    void extend(const Box &b) { extend(b.lo, b.hi); }
    void extendPoint(const float *p) { extend(p, p); }
};
// halfArea(Vec3fa) = d.x*(d.y + d.z) + d.y*d.z on d = upper - lower (common/math/vec3fa.h:279, bbox.h:111); an empty box gives +inf
float halfArea(const Box &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx*(dy + dz) + dy*dz;
}
float area(const Box &b) { return 2.0f*halfArea(b); }

struct PrimRef { Box b; uint32_t id; };
struct PrimInfo {                      // builders/priminfo.h: the items [begin, end) with the box of their boxes and of their doubled centres
    size_t begin = 0, end = 0;
    Box geom, cent;
    size_t size() const { return end - begin; }
};
void addTo(Box &geom, Box &cent, const PrimRef &p)
{
    geom.extend(p.b);
    const float c2[3] = {p.b.lo[0] + p.b.hi[0], p.b.lo[1] + p.b.hi[1], p.b.lo[2] + p.b.hi[2]};     // center2
    cent.extendPoint(c2);
}

// BinMapping + BinSplit (heuristic_binning.h:29-156)
struct Split {
    float sah = INF;
    int dim = -1, pos = 0;
    size_t num = 0;
    float ofs[3] = {0, 0, 0}, scale[3] = {0, 0, 0};
    bool valid() const { return dim != -1; }
    int bin(const PrimRef &p, int d) const { return (int)std::floor(((p.b.lo[d] + p.b.hi[d]) - ofs[d])*scale[d]); }
};

const size_t BINS = 32;                 // NUM_OBJECT_BINS
const int LOG_BLOCK = 2;                // sahBlockSize 4

// HeuristicArrayBinningSAH::sequential_find -> BinInfo::bin + BinInfo::best (heuristic_binning.h:213-372)
Split findSplit(const std::vector<PrimRef> &prims, const PrimInfo &pinfo)
{
    Split s;
    s.num = std::min(BINS, size_t(4.0f + 0.05f*pinfo.size()));
    for (int d = 0; d < 3; ++d) {
        const float diag = pinfo.cent.hi[d] - pinfo.cent.lo[d];
        s.scale[d] = diag > 1E-34f ? (0.99f*s.num)/diag : 0.0f;
        s.ofs[d] = pinfo.cent.lo[d];
    }
    Box bounds[BINS][3];
    int counts[BINS][3];
    for (size_t i = 0; i < BINS; ++i)
        for (int d = 0; d < 3; ++d) { bounds[i][d].clear(); counts[i][d] = 0; }
    for (size_t i = pinfo.begin; i < pinfo.end; ++i)
        for (int d = 0; d < 3; ++d) {
            const int b = s.bin(prims[i], d);
            bounds[b][d].extend(prims[i].b);
            counts[b][d]++;
        }
    // sweep from right to left: merged bounds and counts of the bins i .. num-1
    float rAreas[BINS][3];
    int rCounts[BINS][3];
    {
        int count[3] = {0, 0, 0};
        Box bx[3];
        for (int d = 0; d < 3; ++d) bx[d].clear();
        for (size_t i = s.num - 1; i > 0; --i)
            for (int d = 0; d < 3; ++d) {
                count[d] += counts[i][d];
                rCounts[i][d] = count[d];
                bx[d].extend(bounds[i][d]);
                rAreas[i][d] = halfArea(bx[d]);
            }
    }
    // sweep from left to right: the SAH of splitting in front of bin i (areas x counts in blocks of four items); an empty side is inf*0 = NaN, never best
    float bestSAH[3] = {INF, INF, INF};
    int bestPos[3] = {0, 0, 0};
    {
        const int blocksAdd = (1 << LOG_BLOCK) - 1;
        int count[3] = {0, 0, 0};
        Box bx[3];
        for (int d = 0; d < 3; ++d) bx[d].clear();
        for (size_t i = 1; i < s.num; ++i)
            for (int d = 0; d < 3; ++d) {
                count[d] += counts[i - 1][d];
                bx[d].extend(bounds[i - 1][d]);
                const float lArea = halfArea(bx[d]), rArea = rAreas[i][d];
                const int lCount = (count[d] + blocksAdd) >> LOG_BLOCK, rCount = (rCounts[i][d] + blocksAdd) >> LOG_BLOCK;
                const float sah = lArea*float(lCount) + rArea*float(rCount);
                if (sah < bestSAH[d]) { bestPos[d] = int(i); bestSAH[d] = sah; }
            }
    }
    for (int d = 0; d < 3; ++d) {
        if (s.scale[d] == 0.0f) continue;                      // zero-sized dimension
        if (bestSAH[d] < s.sah && bestPos[d] != 0) { s.dim = d; s.pos = bestPos[d]; s.sah = bestSAH[d]; }
    }
    return s;
}

// HeuristicArrayBinningSAH::sequential_split (heuristic_binning_array_aligned.h:127-166) with serial_partitioning (algorithms/parallel_partition.h:
// 25-65) -- in place, in Embree's order, so that every later min / max sees its operands in Embree's order -- or, without a valid split,
// deterministic_order + splitFallback (:215-249): sorted by item, cut in the middle
void splitSet(std::vector<PrimRef> &prims, const Split &split, const PrimInfo &pinfo, PrimInfo &left, PrimInfo &right)
{
    const size_t begin = pinfo.begin, end = pinfo.end;
    left.geom.clear(); left.cent.clear(); right.geom.clear(); right.cent.clear();
    size_t center;
    if (!split.valid()) {
        std::sort(prims.begin() + begin, prims.begin() + end, [](const PrimRef &a, const PrimRef &b) { return a.id < b.id; });
        center = (begin + end)/2;
        for (size_t i = begin; i < center; ++i) addTo(left.geom, left.cent, prims[i]);
        for (size_t i = center; i < end; ++i) addTo(right.geom, right.cent, prims[i]);
    } else {
        auto isLeft = [&](const PrimRef &p) { return split.bin(p, split.dim) < split.pos; };
        long l = long(begin), r = long(end) - 1;
        for (;;) {
            while (l <= r && isLeft(prims[l])) { addTo(left.geom, left.cent, prims[l]); ++l; }
            while (l <= r && !isLeft(prims[r])) { addTo(right.geom, right.cent, prims[r]); --r; }
            if (r < l) break;
            addTo(left.geom, left.cent, prims[r]);
            addTo(right.geom, right.cent, prims[l]);
            std::swap(prims[l], prims[r]);
            ++l; --r;
        }
        center = size_t(l);
    }
    left.begin = begin; left.end = center;
    right.begin = center; right.end = end;
}

struct Record { PrimInfo pinfo; Split split; };

// GeneralBVHBuilder::recurse (bvh_builder_sah.h:170-282) with minLeafSize = maxLeafSize = 1: a record of one item is a leaf, anything larger
// is split -- always the child with the largest box area next, until the node has four children or only single items --, the children are
// sorted by size (std::sort on <= 4 records is libstdc++'s insertion sort: stable) and stored in that order with their boxes.
int32_t recurse(std::vector<PrimRef> &prims, const Record &current, std::vector<TgHipTopNode> &nodes)
{
    if (current.pinfo.size() <= 1)
        return ~int32_t(prims[current.pinfo.begin].id);
    Record children[4];
    children[0] = current;
    size_t numChildren = 1;
    do {
        float bestArea = -INF;
        long bestChild = -1;
        for (size_t i = 0; i < numChildren; ++i) {
            if (children[i].pinfo.size() <= 1) continue;
            if (area(children[i].pinfo.geom) > bestArea) { bestChild = long(i); bestArea = area(children[i].pinfo.geom); }
        }
        if (bestChild == -1) break;
        Record l, r;
        splitSet(prims, children[bestChild].split, children[bestChild].pinfo, l.pinfo, r.pinfo);
        l.split = findSplit(prims, l.pinfo);
        r.split = findSplit(prims, r.pinfo);
        children[bestChild] = l;
        children[numChildren] = r;
        numChildren++;
    } while (numChildren < 4);
    for (size_t i = 1; i < numChildren; ++i) {                 // insertion sort by std::greater: larger records first, equal sizes keep their order
        Record v = children[i];
        size_t j = i;
        while (j > 0 && v.pinfo.size() > children[j - 1].pinfo.size()) { children[j] = children[j - 1]; --j; }
        children[j] = v;
    }
    const size_t self = nodes.size();
    nodes.emplace_back();
    for (int i = 0; i < 4; ++i) {
        nodes[self].child[i] = TGHIP_TOP_EMPTY;
        for (int k = 0; k < 3; ++k) { nodes[self].lower[i][k] = INF; nodes[self].upper[i][k] = -INF; }     // Node::clear()
    }
    for (size_t i = 0; i < numChildren; ++i)
        for (int k = 0; k < 3; ++k) { nodes[self].lower[i][k] = children[i].pinfo.geom.lo[k]; nodes[self].upper[i][k] = children[i].pinfo.geom.hi[k]; }
    for (size_t i = 0; i < numChildren; ++i) {                  // (Embree recurses last child first; the order does not change the tree)
        const int32_t c = recurse(prims, children[i], nodes);
        nodes[self].child[i] = c;
    }
    return int32_t(self);
}

}

std::vector<TgHipTopNode> buildEmbreeTopTree(const std::vector<TopBox> &boxes)
{
    std::vector<TgHipTopNode> nodes;
    if (boxes.size() < 2)
        return nodes;
    // An item whose box is not valid is left out -- with its id kept for the others -- as Embree does: the builder's primref pass skips it
    // (kernels/common/primref_gen: `if (!mesh->valid(j)) continue`), so no leaf is created and no ray ever reaches that primitive.
    std::vector<PrimRef> prims;
    prims.reserve(boxes.size());
    Record root;
    root.pinfo.geom.clear(); root.pinfo.cent.clear();
    for (size_t i = 0; i < boxes.size(); ++i) {
        bool valid = true;
        PrimRef p;
        for (int k = 0; k < 3; ++k) {
            const float lo = boxes[i].lo[k], hi = boxes[i].hi[k];
            // AccelSet::valid -> isvalid(bounds) (kernels/common/accelset.h, common/math/bbox.h): lower <= upper, both within +-FLT_LARGE (1.844E18)
            if (!(lo <= hi) || !(lo > -1.844E18f && hi < 1.844E18f))
                valid = false;
            p.b.lo[k] = lo; p.b.hi[k] = hi;
        }
        if (!valid)
            continue;
        p.id = uint32_t(i);
        prims.push_back(p);
        addTo(root.pinfo.geom, root.pinfo.cent, p);
    }
    // (fewer than two valid items: no tree -- the device walks the plain record list, which differs from Embree only in that the left-out
    // primitive is still tested; its bounds are NaN or beyond 1.8e18, so is its geometry)
    if (prims.size() < 2)
        return nodes;
    root.pinfo.begin = 0; root.pinfo.end = prims.size();
    root.split = findSplit(prims, root.pinfo);
    recurse(prims, root, nodes);
    return nodes;
}

std::vector<TgHipTopNode> buildSceneTopTree(const TgHipObject *objects, uint32_t numObjects, const TgHipPrimRec *recs, uint32_t numRecs)
{
    std::vector<TgHipTopNode> none;
    if (numRecs < 2 || numRecs > TGHIP_FLAT_MAX_RECS)
        return none;
    std::vector<int> recOf(numObjects, -1);
    for (uint32_t r = 0; r < numRecs; ++r) {
        const uint32_t obj = TGHIP_REC_OBJECT(recs[r].meta);
        if (obj >= numObjects || recOf[obj] >= 0)
            return none;                                         // (two records of one object: a mesh)
        recOf[obj] = int(r);
    }
    std::vector<TopBox> boxes;
    std::vector<uint32_t> itemRec;
    for (uint32_t o = 0; o < numObjects; ++o) {
        if (recOf[o] < 0) continue;                              // infinite and Dirac primitives are not in _finites
        TopBox b;
        if (!referenceLeafBounds(objects[o], TGHIP_REC_KIND(recs[recOf[o]].meta), b.lo, b.hi))
            return none;
        boxes.push_back(b);
        itemRec.push_back(uint32_t(recOf[o]));
    }
    std::vector<TgHipTopNode> nodes = buildEmbreeTopTree(boxes);
    size_t leaves = 0;
    for (TgHipTopNode &n : nodes)
        for (int i = 0; i < 4; ++i)
            if (n.child[i] < 0) {
                n.child[i] = ~int32_t(itemRec[size_t(~n.child[i])]);
                ++leaves;
            }
    if (!nodes.empty() && leaves != boxes.size()) {
        // an item with invalid bounds (NaN, or beyond 1.8e18): Embree leaves it out of its tree, so the reference never intersects it.  The
        // device's ordered walk wants every record in one leaf (tghip_upload_scene), so such a scene is walked as a plain list -- said aloud,
        // because coincident faces may then resolve differently from the reference
        std::fprintf(stderr, "path_tracer_hip: %zu of %zu primitives have invalid bounds; the scene is walked as a plain list, not in the reference's tree order\n",
                     boxes.size() - leaves, boxes.size());
        return none;
    }
    return nodes;
}

bool referenceLeafBounds(const TgHipObject &o, uint32_t kind, float lo[3], float hi[3])
{
    float p[8][3];
    int n = 0;
    switch (kind) {
    case TGHIP_REC_QUAD:
        for (int k = 0; k < 3; ++k) {
            p[0][k] = o.base[k]; p[1][k] = o.base[k] + o.edge0[k]; p[2][k] = o.base[k] + o.edge1[k]; p[3][k] = (o.base[k] + o.edge0[k]) + o.edge1[k];
        }
        n = 4;
        break;
    case TGHIP_REC_CUBE:
        for (int c = 0; c < 8; ++c) {
            const float x = (c & 1) ? o.scale[0] : -o.scale[0], y = (c & 2) ? o.scale[1] : -o.scale[1], z = (c & 4) ? o.scale[2] : -o.scale[2];
            for (int k = 0; k < 3; ++k)
                p[c][k] = o.pos[k] + (o.rot[3*k]*x + o.rot[3*k + 1]*y + o.rot[3*k + 2]*z);
        }
        n = 8;
        break;
    case TGHIP_REC_SPHERE:
        for (int k = 0; k < 3; ++k) { p[0][k] = o.pos[k] - o.scale[0]; p[1][k] = o.pos[k] + o.scale[0]; }
        n = 2;
        break;
    case TGHIP_REC_DISK:                                        // Disk::bounds (Disk.cpp:298-306): pos = _center, edge0 / edge1 = _frame.tangent / bitangent, scale[0] = _r
        for (int k = 0; k < 3; ++k) {
            const float t = o.edge0[k]*o.scale[0], b = o.edge1[k]*o.scale[0];
            p[0][k] = (o.pos[k] - t) - b; p[1][k] = (o.pos[k] + t) - b; p[2][k] = (o.pos[k] + t) + b; p[3][k] = (o.pos[k] - t) + b;
        }
        n = 4;
        break;
    case TGHIP_REC_CYLINDER:                                    // Cylinder::bounds (Cylinder.cpp:272-279): the axis' end points, grown by the radius;
        for (int k = 0; k < 3; ++k) {                           //   _axis = _transform.up().normalized() = the second column of _rot (Mat4f.cpp:40-47), scale = {_radius, _halfHeight, .}
            const float a = o.rot[3*k + 1]*o.scale[1];
            p[0][k] = o.pos[k] + a; p[1][k] = o.pos[k] - a;
        }
        n = 2;
        break;
    default:
        return false;
    }
    for (int k = 0; k < 3; ++k) {                               // Box::grow: min(_min, p) keeps _min unless p < _min (math/MathUtil.hpp:33-52)
        lo[k] = hi[k] = p[0][k];
        for (int c = 1; c < n; ++c) { lo[k] = p[c][k] < lo[k] ? p[c][k] : lo[k]; hi[k] = p[c][k] > hi[k] ? p[c][k] : hi[k]; }
        if (kind == TGHIP_REC_CYLINDER) { lo[k] -= o.scale[0]; hi[k] += o.scale[0]; }       // Box::grow(float)
    }
    return true;
}

}
