#include "WideBvh.hpp"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <functional>
#include <limits>

namespace tungsten_amd {

namespace {

struct Item { int32_t ref; Box3f box; };

float boxArea(const Box3f &b) { return b.empty() ? 0.0f : b.area(); }

float planeSpacing(uint8_t e)            // 2^(e - 127), e in 1..254
{
    uint32_t bits = uint32_t(e) << 23;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// Smallest spacing 2^(e - 127) at which `extent` spans at most 255 steps.
uint8_t spacingExponent(float extent)
{
    if (!(extent > 0.0f))
        return 1;
    int e;
    std::frexp(extent/255.0f, &e);       // extent/255 = m * 2^e, 0.5 <= m < 1  =>  2^e >= extent/255
    int biased = e + 127;
    return uint8_t(std::min(std::max(biased, 1), 254));
}

} // namespace

WideBvhResult buildWideBvh(std::vector<TgHipBvhNode> &bvh2, const std::vector<Box3f> &recBounds)
{
    WideBvhResult out;
    if (bvh2.empty() || recBounds.empty())
        return out;
    const size_t numRecs = recBounds.size();
    // ---- which BVH2 nodes become wide nodes, which subtrees leaves: the SAH-optimal collapse of Ylitie et al. (section 4.1)
    // by dynamic programming over the BVH2, bottom up (children follow their parent in `bvh2`).  cost[n][i] = cheapest way to
    // represent the subtree of n by at most i + 1 roots (wide nodes or leaves) hanging in some parent:
    //   one root: a leaf (<= TGHIP_WIDE_MAX_LEAF records; area x records x PrimCost) or a wide node (area x NodeCost + the
    //             cheapest split of its eight slots between the two BVH2 children);
    //   more:     split the roots between the children, or do not use them all.
    const size_t n2 = bvh2.size();
    const float NodeCost = 1.0f;
    float PrimCost = 0.5f;
    if (const char *e = std::getenv("TGH_WIDE_PRIM_COST")) PrimCost = float(std::atof(e));
    struct Dp { float cost[7]; uint8_t left[7]; uint8_t left8; bool leaf; };    // left[i]: roots given to child0 (0: same as i - 1)
    std::vector<Dp> dp(n2);
    std::vector<float> area(n2);
    std::vector<uint32_t> count(n2), firstRec(n2);
    auto refCount = [&](int32_t ref) { return ref < 0 ? TGHIP_LEAF_COUNT(ref) : count[size_t(ref)]; };
    auto refFirst = [&](int32_t ref) { return ref < 0 ? TGHIP_LEAF_FIRST(ref) : firstRec[size_t(ref)]; };
    auto refCost = [&](int32_t ref, float refArea, int i) {       // i + 1 roots available
        if (ref < 0) return TGHIP_LEAF_COUNT(ref) == 0 ? 0.0f : refArea*float(TGHIP_LEAF_COUNT(ref))*PrimCost;
        return dp[size_t(ref)].cost[i];
    };
    for (size_t k = n2; k-- > 0;) {
        const TgHipBvhNode &n = bvh2[k];
        Box3f b0, b1;
        b0.lo = Vec3f(n.lo0[0], n.lo0[1], n.lo0[2]); b0.hi = Vec3f(n.hi0[0], n.hi0[1], n.hi0[2]);
        b1.lo = Vec3f(n.lo1[0], n.lo1[1], n.lo1[2]); b1.hi = Vec3f(n.hi1[0], n.hi1[1], n.hi1[2]);
        const bool e0 = n.child0 < 0 && TGHIP_LEAF_COUNT(n.child0) == 0, e1 = n.child1 < 0 && TGHIP_LEAF_COUNT(n.child1) == 0;
        if ((n.child0 >= 0 && size_t(n.child0) <= k) || (n.child1 >= 0 && size_t(n.child1) <= k)) return out;   // not in pre-order: no wide BVH
        Box3f box;
        if (!e0) box.grow(b0);
        if (!e1) box.grow(b1);
        area[k] = boxArea(box);
        count[k] = refCount(n.child0) + refCount(n.child1);
        firstRec[k] = e0 ? refFirst(n.child1) : refFirst(n.child0);
        const float a0 = boxArea(b0), a1 = boxArea(b1);
        Dp &d = dp[k];
        // a wide node here: its eight slots split between the two children
        float best8 = std::numeric_limits<float>::infinity();
        d.left8 = 1;
        for (int l = 1; l <= 7; ++l) {
            const float c = refCost(n.child0, a0, l - 1) + refCost(n.child1, a1, 7 - l);
            if (c < best8) { best8 = c; d.left8 = uint8_t(l); }
        }
        const float asNode = area[k]*NodeCost + best8;
        // the two children of `n` must stay apart when they are not one run of records (child1's records follow child0's in a
        // BVH2 built depth first; an empty child leaves one run)
        const bool oneRun = e0 || e1 || refFirst(n.child0) + refCount(n.child0) == refFirst(n.child1);
        const float asLeaf = (count[k] <= TGHIP_WIDE_MAX_LEAF && count[k] > 0 && oneRun) ? area[k]*float(count[k])*PrimCost : std::numeric_limits<float>::infinity();
        d.leaf = asLeaf <= asNode;
        d.cost[0] = std::min(asLeaf, asNode);
        d.left[0] = 0;
        for (int i = 1; i < 7; ++i) {            // i + 1 roots
            d.cost[i] = d.cost[i - 1];
            d.left[i] = 0;
            for (int l = 1; l <= i; ++l) {       // l roots to child0, i + 1 - l to child1
                const float c = refCost(n.child0, a0, l - 1) + refCost(n.child1, a1, i - l);
                if (c < d.cost[i]) { d.cost[i] = c; d.left[i] = uint8_t(l); }
            }
        }
    }
    // the roots representing `ref` (box `refBox`) when it may use up to `roots` of them
    std::function<void(int32_t, const Box3f &, int, std::vector<Item> &)> collect = [&](int32_t ref, const Box3f &refBox, int roots, std::vector<Item> &dst) {
        if (ref < 0) {
            if (TGHIP_LEAF_COUNT(ref) > 0) dst.push_back({ref, refBox});
            return;
        }
        const Dp &d = dp[size_t(ref)];
        int i = roots - 1;
        while (i > 0 && d.left[i] == 0) --i;     // "do not use them all"
        if (i == 0) {
            if (d.leaf) dst.push_back({TGHIP_MAKE_LEAF(firstRec[size_t(ref)], count[size_t(ref)]), refBox});
            else        dst.push_back({ref, refBox});
            return;
        }
        const TgHipBvhNode &n = bvh2[size_t(ref)];
        Box3f b0, b1;
        b0.lo = Vec3f(n.lo0[0], n.lo0[1], n.lo0[2]); b0.hi = Vec3f(n.hi0[0], n.hi0[1], n.hi0[2]);
        b1.lo = Vec3f(n.lo1[0], n.lo1[1], n.lo1[2]); b1.hi = Vec3f(n.hi1[0], n.hi1[1], n.hi1[2]);
        collect(n.child0, b0, d.left[i], dst);
        collect(n.child1, b1, i + 1 - d.left[i], dst);
    };

    struct Pending { int32_t bvh2Node; int depth; };
    std::deque<Pending> queue;
    out.nodes.emplace_back();
    queue.push_back({0, 1});
    out.order.reserve(numRecs);
    std::vector<uint32_t> newFirst(numRecs, 0xFFFFFFFFu);   // old position of a record -> its new position
    size_t next = 0;                                        // wide node being filled (breadth first = allocation order)

    while (!queue.empty()) {
        const Pending cur = queue.front();
        queue.pop_front();
        const size_t self = next++;
        out.depth = std::max(out.depth, cur.depth);
        if (cur.depth > TGHIP_MAX_WIDE_DEPTH) { out.nodes.clear(); out.order.clear(); return out; }

        // ---- the node's children: the DP's split of the eight slots between the BVH2 node's two children
        std::vector<Item> items;
        {
            const TgHipBvhNode &n = bvh2[size_t(cur.bvh2Node)];
            Box3f b0, b1;
            b0.lo = Vec3f(n.lo0[0], n.lo0[1], n.lo0[2]); b0.hi = Vec3f(n.hi0[0], n.hi0[1], n.hi0[2]);
            b1.lo = Vec3f(n.lo1[0], n.lo1[1], n.lo1[2]); b1.hi = Vec3f(n.hi1[0], n.hi1[1], n.hi1[2]);
            const int l = dp[size_t(cur.bvh2Node)].left8;
            collect(n.child0, b0, l, items);
            collect(n.child1, b1, 8 - l, items);
        }
        Box3f bounds;
        for (const Item &it : items) {
            if (it.ref < 0 && TGHIP_LEAF_COUNT(it.ref) > TGHIP_WIDE_MAX_LEAF) { out.nodes.clear(); out.order.clear(); return out; }
            bounds.grow(it.box);
        }
        if (items.empty() || items.size() > 8) {
            if (items.size() > 8) { out.nodes.clear(); out.order.clear(); return out; }
            bounds.lo = bounds.hi = Vec3f(0.0f);  // (an empty scene's root)
        }

        // ---- slots: greedy assignment of children to the octant their centre lies in, as seen from the node's centre
        const Vec3f centre = (bounds.lo + bounds.hi)*0.5f;
        int slotOf[8], itemAt[8];
        for (int i = 0; i < 8; ++i) { slotOf[i] = -1; itemAt[i] = -1; }
        for (size_t round = 0; round < items.size(); ++round) {
            float bestCost = -std::numeric_limits<float>::infinity();
            int bi = -1, bs = -1;
            for (size_t i = 0; i < items.size(); ++i) {
                if (slotOf[i] >= 0) continue;
                const Vec3f d = (items[i].box.lo + items[i].box.hi)*0.5f - centre;
                for (int s = 0; s < 8; ++s) {
                    if (itemAt[s] >= 0) continue;
                    const float cost = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
                    if (cost > bestCost) { bestCost = cost; bi = int(i); bs = s; }
                }
            }
            slotOf[bi] = bs;
            itemAt[bs] = bi;
        }

        // ---- quantisation: lower planes round down, upper planes up -- with SLACK.  The walk computes a plane's distance as
        // fma(q, spacing/d, (origin - o)/d) (pt_kernels.h: wideVisit): when the node's origin is far from the ray's compared with the plane's
        // distance the two terms cancel, and the result is off by ulps of |origin - o|/d, not of the distance.  A quantised plane that
        // coincides with its child's box -- a child on the node's own boundary, a zero-thickness axis-aligned quad on a grid plane -- would
        // then be culled for grazing rays.  So the grid starts one step below the node's box, and every plane keeps at least 1/64 of a
        // step between itself and the box it bounds: in the large nodes near the root -- the only ones where the cancellation exceeds what
        // wideVisit's relative padding of the far distance covers -- that is 16 to 250 times the error (2^-22 of the scene's size against
        // a step of 1/255 of the node's); a quarter step cost 5 % more node visits, 1/64 costs under 1 %.
        TgHipWideNode node;
        std::memset(&node, 0, sizeof(node));
        const float minSlack = 1.0f/64.0f;
        for (int a = 0; a < 3; ++a) {
            uint8_t e = spacingExponent(bounds.hi[a] - bounds.lo[a]);
            for (;;) {
                const float sp = planeSpacing(e);
                const float org = bounds.lo[a] - sp;
                bool fits = true;
                for (int s = 0; s < 8 && fits; ++s) {
                    if (itemAt[s] < 0) { node.qlo[a][s] = 255; node.qhi[a][s] = 0; continue; }
                    const Box3f &b = items[size_t(itemAt[s])].box;
                    float ql = std::floor((b.lo[a] - org)/sp), qh = std::ceil((b.hi[a] - org)/sp);
                    ql = std::min(std::max(ql, 0.0f), 255.0f);
                    qh = std::max(qh, 0.0f);
                    while (ql > 0.0f && org + ql*sp > b.lo[a]) ql -= 1.0f;
                    while (qh <= 255.0f && org + qh*sp < b.hi[a]) qh += 1.0f;
                    if (ql > 0.0f && b.lo[a] - (org + ql*sp) < minSlack*sp) ql -= 1.0f;
                    if (qh <= 255.0f && (org + qh*sp) - b.hi[a] < minSlack*sp) qh += 1.0f;
                    if (qh > 255.0f) { fits = false; break; }
                    node.qlo[a][s] = uint8_t(ql);
                    node.qhi[a][s] = uint8_t(qh);
                }
                node.origin[a] = org;
                if (fits || e >= 254)
                    break;
                ++e;
            }
            node.exp[a] = e;
        }

        // ---- children: internal ones become consecutive nodes, the leaves' records one contiguous run
        node.child_base = uint32_t(out.nodes.size());
        node.rec_base = uint32_t(out.order.size());
        for (int s = 0; s < 8; ++s) {
            if (itemAt[s] < 0) continue;
            const Item &it = items[size_t(itemAt[s])];
            if (it.ref >= 0) {
                node.imask |= uint8_t(1u << s);
                out.nodes.emplace_back();
                queue.push_back({it.ref, cur.depth + 1});
            } else {
                const uint32_t first = TGHIP_LEAF_FIRST(it.ref), count = TGHIP_LEAF_COUNT(it.ref);
                node.leaf_valid |= ((1u << count) - 1u) << (4*s);
                for (uint32_t r = first; r < first + count; ++r) {
                    newFirst[r] = uint32_t(out.order.size());
                    out.order.push_back(r);
                }
            }
        }
        out.nodes[self] = node;
    }
    if (out.order.size() != numRecs) { out.nodes.clear(); out.order.clear(); return out; }

    // the BVH2's leaves follow their records
    for (TgHipBvhNode &n : bvh2) {
        if (n.child0 < 0 && TGHIP_LEAF_COUNT(n.child0) > 0) n.child0 = TGHIP_MAKE_LEAF(newFirst[TGHIP_LEAF_FIRST(n.child0)], TGHIP_LEAF_COUNT(n.child0));
        if (n.child1 < 0 && TGHIP_LEAF_COUNT(n.child1) > 0) n.child1 = TGHIP_MAKE_LEAF(newFirst[TGHIP_LEAF_FIRST(n.child1)], TGHIP_LEAF_COUNT(n.child1));
    }
    return out;
}

} // namespace tungsten_amd
