#include "BvhBuilder.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace tungsten_amd {

namespace {

const int NumBins = 32;
// SAH constants; TGH_BVH_TRAV_COST / TGH_BVH_MAX_LEAF override them for tuning experiments (profiles/README.md)
static float TraversalCost = 1.0f;
const float IntersectionCost = 1.0f;

struct Ref { Box3f box; Vec3f centroid; uint32_t prim; };

struct Builder
{
    std::vector<Ref> refs;
    std::vector<TgHipBvhNode> nodes;
    int maxLeaf;
    int maxDepth = 0;
    int maxLeafSeen = 0;
    double sah = 0.0;
    float rootArea = 1.0f;

    static float safeArea(const Box3f &b) { return b.empty() ? 0.0f : b.area(); }

    // Picks a split of refs[begin, end); returns the partition point (begin < mid < end) or -1
    // when a leaf is cheaper / no split separates the centroids.
    int findSplit(int begin, int end, const Box3f &bounds, const Box3f &centBounds, bool mustSplit)
    {
        int n = end - begin;
        Vec3f diag = centBounds.hi - centBounds.lo;
        float bestCost = std::numeric_limits<float>::infinity();
        int bestAxis = -1, bestBin = -1;

        for (int axis = 0; axis < 3; ++axis) {
            if (!(diag[axis] > 0.0f))
                continue;
            Box3f binBox[NumBins];
            int binCount[NumBins] = {0};
            float scale = NumBins/diag[axis];
            for (int i = begin; i < end; ++i) {
                int b = std::min(NumBins - 1, std::max(0, int((refs[i].centroid[axis] - centBounds.lo[axis])*scale)));
                binBox[b].grow(refs[i].box);
                binCount[b]++;
            }
            float rightArea[NumBins];
            int rightCount[NumBins];
            Box3f acc;
            int cnt = 0;
            for (int b = NumBins - 1; b > 0; --b) {
                acc.grow(binBox[b]);
                cnt += binCount[b];
                rightArea[b] = safeArea(acc);
                rightCount[b] = cnt;
            }
            acc = Box3f();
            cnt = 0;
            for (int b = 0; b < NumBins - 1; ++b) {
                acc.grow(binBox[b]);
                cnt += binCount[b];
                if (cnt == 0 || rightCount[b + 1] == 0)
                    continue;
                float cost = safeArea(acc)*cnt + rightArea[b + 1]*rightCount[b + 1];
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestBin = b;
                }
            }
        }

        float parentArea = std::max(safeArea(bounds), 1e-30f);
        float splitCost = TraversalCost + IntersectionCost*bestCost/parentArea;
        float leafCost = IntersectionCost*n;
        if (bestAxis < 0) {
            if (!mustSplit)
                return -1;
            return begin + n/2;    // identical centroids: arbitrary halving keeps leaves bounded
        }
        if (!mustSplit && leafCost <= splitCost)
            return -1;

        float scale = NumBins/diag[bestAxis];
        float lo = centBounds.lo[bestAxis];
        auto midIt = std::partition(refs.begin() + begin, refs.begin() + end, [&](const Ref &r) {
            int b = std::min(NumBins - 1, std::max(0, int((r.centroid[bestAxis] - lo)*scale)));
            return b <= bestBin;
        });
        int mid = int(midIt - refs.begin());
        if (mid == begin || mid == end)
            mid = begin + n/2;
        return mid;
    }

    int medianSplit(int begin, int end, const Box3f &centBounds)
    {
        Vec3f diag = centBounds.hi - centBounds.lo;
        int axis = diag[0] > diag[1] ? (diag[0] > diag[2] ? 0 : 2) : (diag[1] > diag[2] ? 1 : 2);
        int mid = begin + (end - begin)/2;
        std::nth_element(refs.begin() + begin, refs.begin() + mid, refs.begin() + end,
                         [axis](const Ref &a, const Ref &b) { return a.centroid[axis] < b.centroid[axis]; });
        return mid;
    }

    // Returns the child reference for refs[begin, end) and its bounds.
    int32_t build(int begin, int end, int depth, Box3f &outBounds)
    {
        Box3f bounds, centBounds;
        for (int i = begin; i < end; ++i) {
            bounds.grow(refs[i].box);
            centBounds.grow(refs[i].centroid);
        }
        outBounds = bounds;
        int n = end - begin;
        maxDepth = std::max(maxDepth, depth);

        bool depthExhausted = depth >= TGHIP_MAX_TREE_DEPTH - 2;
        int mid = -1;
        if (n > maxLeaf || (n > 1 && !depthExhausted)) {
            if (depthExhausted && n <= TGHIP_MAX_LEAF) {
                mid = -1;
            } else if (depth >= TGHIP_MAX_TREE_DEPTH - 10) {
                mid = medianSplit(begin, end, centBounds);   // guarantees log2 termination below the cap
            } else {
                mid = findSplit(begin, end, bounds, centBounds, n > maxLeaf);
            }
        }
        if (mid < 0) {
            maxLeafSeen = std::max(maxLeafSeen, n);
            sah += IntersectionCost*n*safeArea(bounds)/rootArea;
            return TGHIP_MAKE_LEAF(begin, n);
        }

        int32_t nodeIdx = int32_t(nodes.size());
        nodes.emplace_back();
        sah += TraversalCost*safeArea(bounds)/rootArea;
        Box3f b0, b1;
        int32_t c0 = build(begin, mid, depth + 1, b0);
        int32_t c1 = build(mid, end, depth + 1, b1);
        TgHipBvhNode &node = nodes[size_t(nodeIdx)];
        for (int k = 0; k < 3; ++k) {
            node.lo0[k] = b0.lo[k]; node.hi0[k] = b0.hi[k];
            node.lo1[k] = b1.lo[k]; node.hi1[k] = b1.hi[k];
        }
        node.child0 = c0;
        node.child1 = c1;
        node.pad[0] = node.pad[1] = 0;
        return nodeIdx;
    }
};

} // namespace

BvhBuildResult buildBvh(const std::vector<Box3f> &primBounds, int maxLeafSize)
{
    if (const char *e = std::getenv("TGH_BVH_TRAV_COST")) TraversalCost = float(std::atof(e));
    if (const char *e = std::getenv("TGH_BVH_MAX_LEAF")) maxLeafSize = std::atoi(e);
    Builder b;
    b.maxLeaf = std::min(std::max(maxLeafSize, 1), int(TGHIP_MAX_LEAF));
    b.refs.resize(primBounds.size());
    Box3f all;
    for (size_t i = 0; i < primBounds.size(); ++i) {
        b.refs[i].box = primBounds[i];
        b.refs[i].centroid = (primBounds[i].lo + primBounds[i].hi)*0.5f;
        b.refs[i].prim = uint32_t(i);
        all.grow(primBounds[i]);
    }
    b.rootArea = std::max(Builder::safeArea(all), 1e-30f);
    b.nodes.reserve(primBounds.size());

    BvhBuildResult result;
    const float inf = std::numeric_limits<float>::infinity();
    if (primBounds.empty()) {
        TgHipBvhNode root;
        std::memset(&root, 0, sizeof(root));
        for (int k = 0; k < 3; ++k) { root.lo0[k] = root.lo1[k] = inf; root.hi0[k] = root.hi1[k] = -inf; }
        root.child0 = root.child1 = TGHIP_MAKE_LEAF(0, 0);
        result.nodes.push_back(root);
        return result;
    }

    Box3f rootBounds;
    int32_t rootRef = b.build(0, int(b.refs.size()), 0, rootBounds);
    if (rootRef < 0) {
        // the whole scene fits one leaf: wrap it in a root node whose second child is empty
        TgHipBvhNode root;
        std::memset(&root, 0, sizeof(root));
        for (int k = 0; k < 3; ++k) {
            root.lo0[k] = rootBounds.lo[k]; root.hi0[k] = rootBounds.hi[k];
            root.lo1[k] = inf; root.hi1[k] = -inf;
        }
        root.child0 = rootRef;
        root.child1 = TGHIP_MAKE_LEAF(0, 0);
        b.nodes.push_back(root);
    }
    result.nodes.swap(b.nodes);
    result.order.resize(b.refs.size());
    for (size_t i = 0; i < b.refs.size(); ++i)
        result.order[i] = b.refs[i].prim;
    result.maxDepth = b.maxDepth;
    result.maxLeafSize = b.maxLeafSeen;
    result.sahCost = b.sah;
    return result;
}

} // namespace tungsten_amd
