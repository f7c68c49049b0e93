// oracle/ref_embree_tree.cpp -- TEST INFRASTRUCTURE (build container only).  Our own program against the reference's vendored Embree
// (/root/reference/src/thirdparty/embree, 2.11.0, built by oracle/Makefile.ref with SSE4.2 as its maximum ISA like the reference's CMakeLists.txt:17):
// commits a scene of ONE user geometry with n items the way the reference's TraceableScene does (renderer/TraceableScene.hpp:112-134:
// RTC_SCENE_STATIC | RTC_SCENE_INCOHERENT, RTC_INTERSECT1, rtcNewUserGeometry(scene, n), a bounds callback) and writes out the BVH4 Embree
// built over the items -- the tree csrc/host/EmbreeTopTree.cpp restates (tests/test_top_tree.py compares the two; tools/make_top_tree_golden.py
// writes the committed fixtures with this program).
//
//   ref_embree_tree <boxes.bin> <out.txt>
//       boxes.bin: u32 sets, then per set u32 n and n x 6 float32 (lower xyz, upper xyz)
//       out.txt:   per set one line: the tree in preorder, a node as "N(" + four children + ")", a child as its box (six floats as hex bit
//                  patterns) followed by either a nested node or "L<item>"; an unused slot is "-".  A scene of one item: "L0" (the root is the leaf).
//
// Nothing here is copied from Embree; it reads its public structures (kernels/bvh/bvh.h) through their own accessors.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "embree2/rtcore.h"
#include "kernels/common/scene.h"
#include "kernels/bvh/bvh.h"
#include "kernels/geometry/object.h"

using namespace embree;

struct BoxSet { std::vector<float> b; };

static void boundsFunc(void *ptr, size_t i, RTCBounds &o)
{
    const float *b = static_cast<BoxSet *>(ptr)->b.data() + 6*i;
    o.lower_x = b[0]; o.lower_y = b[1]; o.lower_z = b[2];
    o.upper_x = b[3]; o.upper_y = b[4]; o.upper_z = b[5];
}
static void intersectFunc(void *, RTCRay &, size_t) {}

static void hex(std::string &s, float f)
{
    unsigned u; std::memcpy(&u, &f, 4);
    char buf[16]; std::snprintf(buf, sizeof buf, "%08x ", u);
    s += buf;
}
static void dump(BVH4::NodeRef ref, std::string &s)
{
    if (ref.isLeaf()) {
        size_t num;
        const Object *o = reinterpret_cast<const Object *>(ref.leaf(num));
        if (num != 1) { std::fprintf(stderr, "leaf with %zu items\n", num); std::exit(1); }
        s += "L" + std::to_string(o->primID) + " ";
        return;
    }
    if (!ref.isNode()) { std::fprintf(stderr, "unexpected node type\n"); std::exit(1); }
    const BVH4::Node *n = ref.node();
    s += "N( ";
    for (size_t i = 0; i < 4; ++i) {
        if (n->child(i) == BVH4::emptyNode) { s += "- "; continue; }
        hex(s, n->lower_x[i]); hex(s, n->lower_y[i]); hex(s, n->lower_z[i]);
        hex(s, n->upper_x[i]); hex(s, n->upper_y[i]); hex(s, n->upper_z[i]);
        dump(n->child(i), s);
    }
    s += ") ";
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: ref_embree_tree <boxes.bin> <out.txt>\n"); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    FILE *out = std::fopen(argv[2], "w");
    if (!f || !out) return 2;
    RTCDevice dev = rtcNewDevice(nullptr);                 // EmbreeUtil::initDevice (primitives/EmbreeUtil.cpp:9-12)
    unsigned sets = 0;
    if (std::fread(&sets, 4, 1, f) != 1) return 2;
    for (unsigned k = 0; k < sets; ++k) {
        unsigned n = 0;
        if (std::fread(&n, 4, 1, f) != 1) return 2;
        BoxSet set; set.b.resize(size_t(n)*6);
        if (n && std::fread(set.b.data(), 24, n, f) != n) return 2;
        RTCScene scene = rtcDeviceNewScene(dev, RTC_SCENE_STATIC | RTC_SCENE_INCOHERENT, RTC_INTERSECT1);
        unsigned geom = rtcNewUserGeometry(scene, n);
        rtcSetUserData(scene, geom, &set);
        rtcSetBoundsFunction(scene, geom, boundsFunc);
        rtcSetIntersectFunction(scene, geom, intersectFunc);
        rtcCommit(scene);
        Scene *sc = reinterpret_cast<Scene *>(scene);
        std::string s;
        int found = 0;
        for (size_t a = 0; a < sc->accels.accels.size(); ++a) {
            Accel *acc = sc->accels.accels[a];
            BVH4 *bvh = reinterpret_cast<BVH4 *>(acc->intersectors.ptr);
            if (!bvh || bvh->numPrimitives == 0) continue;
            dump(bvh->root, s);
            found++;
        }
        if (found != 1) { std::fprintf(stderr, "set %u: %d non-empty accels\n", k, found); return 1; }
        std::fprintf(out, "%s\n", s.c_str());
        rtcDeleteScene(scene);
    }
    std::fclose(out);
    std::fclose(f);
    rtcDeleteDevice(dev);
    return 0;
}
