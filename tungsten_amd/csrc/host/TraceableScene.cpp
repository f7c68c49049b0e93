#include "TraceableScene.hpp"
#include "EmbreeTopTree.hpp"
#include "Sampling.hpp"
#include "BvhBuilder.hpp"
#include "WideBvh.hpp"
#include "RefInstanceBvh.hpp"
#include "Integrator.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>

namespace tungsten_amd {

TraceableScene::TraceableScene(Scene &scene, Integrator *integrator, uint32_t seed)
: _scene(scene), _integrator(integrator), _seed(seed)
{
    std::memset(&_desc, 0, sizeof(_desc));
    flatten();
    if (_integrator)
        _integrator->prepareForRender(*this, seed);    // TraceableScene.hpp:136
}

TraceableScene::~TraceableScene()
{
    if (_integrator)
        _integrator->teardownAfterRender();            // TraceableScene.hpp:141
}

static void copy3(float *dst, const Vec3f &v) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; }
static void copyRot(float *dst, const Mat4f &m)
{
    dst[0] = m[0]; dst[1] = m[1]; dst[2] = m[2];
    dst[3] = m[4]; dst[4] = m[5]; dst[5] = m[6];
    dst[6] = m[8]; dst[7] = m[9]; dst[8] = m[10];
}

SceneAccel buildSceneAccel(std::vector<TgHipPrimRec> &_recs, std::vector<TgHipTriAttr> &_triAttrs, const std::vector<Box3f> &recBounds, bool haveInstances)
{
    SceneAccel out;
    // ---- BVH ---------------------------------------------------------------------------------
    // Leaf size, measured on MI355X (profiles/README.md): single-record leaves are 5 % faster while the tree fits the
    // 4 MiB-per-XCD L2 (materialtest: 80 K records), leaves of <= 4 are 3 % faster once it does not (1 M records).
    // (instance records always sit alone in their leaf: the traversal enters a master from a leaf and returns to its parent)
    BvhBuildResult bvh = buildBvh(recBounds, (haveInstances || recBounds.size() < (1u << 18)) ? 1 : 4);
    if (bvh.maxDepth > TGHIP_MAX_TREE_DEPTH - 1)
        throw std::runtime_error("BVH deeper than the device traversal stack");
    std::vector<TgHipPrimRec> recs(_recs.size());
    std::vector<TgHipTriAttr> attrs(_recs.size());
    for (size_t i = 0; i < bvh.order.size(); ++i) {
        recs[i] = _recs[bvh.order[i]];
        attrs[i] = _triAttrs[bvh.order[i]];
    }
    _recs.swap(recs);
    _triAttrs.swap(attrs);
    out.order = bvh.order;
    out.nodes.swap(bvh.nodes);
    out.bvhDepth = bvh.maxDepth;
    out.sahCost = bvh.sahCost;
    // ---- the 8-wide BVH the single-level traversal kernels walk (WideBvh.hpp): the BVH2 collapsed, records re-ordered by
    // wide node.  Flat-list scenes are intersected without a tree.  With instances the top-level tree's leaves hold the
    // instance records and every master gets a wide subtree of its own behind it (TraceableScene::flatten).
    const bool wantWide = (_recs.size() > TGHIP_FLAT_MAX_RECS || haveInstances) && !std::getenv("TGH_NO_WIDE_BVH");
    if (wantWide) {
        std::vector<Box3f> ordered(recBounds.size());
        for (size_t i = 0; i < bvh.order.size(); ++i)
            ordered[i] = recBounds[bvh.order[i]];
        WideBvhResult wide = buildWideBvh(out.nodes, ordered);
        if (!wide.nodes.empty()) {
            for (size_t i = 0; i < wide.order.size(); ++i) {
                recs[i] = _recs[wide.order[i]];
                attrs[i] = _triAttrs[wide.order[i]];
            }
            _recs.swap(recs);
            _triAttrs.swap(attrs);
            std::vector<uint32_t> composed(wide.order.size());
            for (size_t i = 0; i < wide.order.size(); ++i)
                composed[i] = out.order[wide.order[i]];
            out.order.swap(composed);
            out.wideNodes.swap(wide.nodes);
            out.wideDepth = wide.depth;
        }
    }
    return out;
}

// Scenes with `instances` primitives (include/tungsten_hip.h, the instance-set record).  `recs` / `attrs` / `recBounds` come in as the K
// non-instance records and go out as the scene's whole record array: [the non-instance and the N instance records in the wide tree's
// order | one set record per `instances` primitive | the masters' records, each master's in its own trees' order].
//  * The wide BVH -- any-hit shadow queries -- is built over the non-instance records and the instance records, every instance boxed
//    tightly (a ray that is to hit an instance passes that box somewhere along its whole length; whether the reference lets it INTO the
//    instance is the leaf test the walk makes at the record, pt_wavefront.h); it decides the record order.
//  * The scene's BVH2 -- closest hits -- is built over the non-instance records and one record per `instances` primitive, behind which
//    the reference's own tree over the instances follows (Instance::prepareForRender, Instance.cpp:392-428: the reference's intersect
//    keeps the LAST hit in that tree's visiting order -- each instance gets a ray with farT = infinity --, so the tree is restated node
//    for node, RefInstanceBvh.cpp, and walked in the reference's order).
//  * Behind the top level: every master's records in master space with a BVH2 subtree and a wide subtree of its own.
// (TraceableScene::flatten and, for the reference-side flattener of oracle/ref_binding, tgh_accel_build_instanced.)
InstancedAccel buildInstancedAccel(std::vector<TgHipPrimRec> &_recs, std::vector<TgHipTriAttr> &_triAttrs, const std::vector<Box3f> &recBounds,
                                   const std::vector<InstanceSetInput> &sets, const std::vector<MasterInput> &masters)
{
    InstancedAccel out;
    std::vector<TgHipBvhNode> &_nodes = out.nodes;
    std::vector<TgHipWideNode> &_wideNodes = out.wideNodes;
    std::vector<uint32_t> &_instPrims = out.instPrims;
    std::vector<float> &_instLeafBoxes = out.instLeafBoxes, &_instTightBoxes = out.instTightBoxes;
    int &_bvhDepth = out.bvhDepth, &_wideDepth = out.wideDepth;

    struct Set { TgHipPrimRec rec; Box3f bounds; size_t firstInst; RefInstanceBvh tree; };
    std::vector<Set> instSets;
    std::vector<TgHipPrimRec> instRecs;
    std::vector<Box3f> instTightBounds;
    for (const InstanceSetInput &in : sets) {
        if (in.recs.empty())
            continue;
        if (in.refBounds.size() != in.recs.size() || in.tightBounds.size() != in.recs.size())
            throw std::runtime_error("instance set: one reference box and one tight box per instance record");
        Set set;
        set.firstInst = instRecs.size();
        for (const TgHipPrimRec &r : in.recs) {
            uint32_t masterSlot;
            std::memcpy(&masterSlot, &r.c[0], 4);
            if (masterSlot >= masters.size() || masters[masterSlot].recs.empty())
                throw std::runtime_error("instance of a master that does not exist or is empty");
        }
        instRecs.insert(instRecs.end(), in.recs.begin(), in.recs.end());
        instTightBounds.insert(instTightBounds.end(), in.tightBounds.begin(), in.tightBounds.end());
        set.tree = buildRefInstanceBvh(in.refBounds);
        set.bounds = set.tree.bounds;
        std::memset(&set.rec, 0, sizeof(set.rec));
        copy3(set.rec.a, set.bounds.lo);
        copy3(set.rec.b, set.bounds.hi);
        set.rec.meta = (uint32_t(TGHIP_REC_INSTANCE_SET) << 29) | in.objMeta;
        // every leaf's box as its parent holds it (a tree that is one leaf: the root's bounds), at the leaf's first slot of inst_prims,
        // and that slot in the leaf's instance records
        const size_t slotBase = _instLeafBoxes.size()/8;         // (= this set's first slot of inst_prims: one box slot per leaf slot)
        _instLeafBoxes.resize(_instLeafBoxes.size() + 8*set.tree.primIndices.size(), 0.0f);
        auto leafBox = [&](uint32_t leafNode, const Box3f &box) {
            const TgHipInstNode &leaf = set.tree.nodes[leafNode];
            const uint32_t slot = uint32_t(slotBase + leaf.left);
            for (int k = 0; k < 3; ++k) { _instLeafBoxes[8*size_t(slot) + k] = box.lo[k]; _instLeafBoxes[8*size_t(slot) + 4 + k] = box.hi[k]; }
            for (uint32_t k = 0; k < leaf.count; ++k) {
                const size_t rec = set.firstInst + set.tree.primIndices[leaf.left + k];
                std::memcpy(&instRecs[rec].c[1], &slot, 4);
            }
        };
        if (set.tree.nodes[0].count != 0) {
            leafBox(0, set.bounds);
        } else {
            for (size_t ni = 0; ni < set.tree.nodes.size(); ++ni) {
                const TgHipInstNode &n = set.tree.nodes[ni];
                if (n.count != 0) continue;
                for (uint32_t c = 0; c < 2; ++c) {
                    if (set.tree.nodes[n.left + c].count == 0) continue;
                    Box3f box;
                    for (int k = 0; k < 3; ++k) { box.lo[k] = n.box[k*4 + c]; box.hi[k] = n.box[k*4 + 2 + c]; }
                    leafBox(n.left + c, box);
                }
            }
        }
        instSets.push_back(std::move(set));
    }
    const uint32_t numInstances = uint32_t(instRecs.size());
    if (numInstances == 0)
        throw std::runtime_error("buildInstancedAccel: no instances");
    int refDepth = 0;
    {
        const size_t K = _recs.size(), N = instRecs.size(), S = instSets.size();
        std::vector<Box3f> wideBounds(recBounds);
        for (size_t i = 0; i < N; ++i) {
            _recs.push_back(instRecs[i]);
            _triAttrs.emplace_back();
            std::memset(&_triAttrs.back(), 0, sizeof(TgHipTriAttr));
            _triAttrs.back().bsdf = -1;
            wideBounds.push_back(instTightBounds[i]);
        }
        SceneAccel accel = buildSceneAccel(_recs, _triAttrs, wideBounds, true);
        _wideNodes.swap(accel.wideNodes);
        _wideDepth = accel.wideDepth;
        out.sahCost = accel.sahCost;
        std::vector<uint32_t> slotOf(K + N);                 // caller's record -> its slot
        for (size_t i = 0; i < accel.order.size(); ++i)
            slotOf[accel.order[i]] = uint32_t(i);
        _instTightBoxes.assign(8*(K + N + S), 0.0f);
        for (size_t i = 0; i < N; ++i)
            for (int k = 0; k < 3; ++k) {
                _instTightBoxes[8*size_t(slotOf[K + i]) + k] = instTightBounds[i].lo[k];
                _instTightBoxes[8*size_t(slotOf[K + i]) + 4 + k] = instTightBounds[i].hi[k];
            }
        for (size_t si = 0; si < S; ++si) {
            _recs.push_back(instSets[si].rec);
            _triAttrs.emplace_back();
            std::memset(&_triAttrs.back(), 0, sizeof(TgHipTriAttr));
            _triAttrs.back().bsdf = -1;
        }
        std::vector<Box3f> sceneBounds(recBounds);
        for (size_t si = 0; si < S; ++si)
            sceneBounds.push_back(instSets[si].bounds);
        BvhBuildResult top = buildBvh(sceneBounds, 1);
        auto finalSlot = [&](uint32_t input) { return input < K ? slotOf[input] : uint32_t(K + N + (input - K)); };
        for (TgHipBvhNode &n : top.nodes)
            for (int32_t *ref : {&n.child0, &n.child1})
                if (*ref < 0)
                    *ref = TGHIP_MAKE_LEAF(finalSlot(top.order[TGHIP_LEAF_FIRST(*ref)]), 1);
        _nodes.swap(top.nodes);
        _bvhDepth = top.maxDepth;
        // the reference's trees as BVH2 nodes with its exact child boxes; their leaf references index inst_prims
        for (size_t si = 0; si < S; ++si) {
            const Set &set = instSets[si];
            const uint32_t primBase = uint32_t(_instPrims.size());
            for (uint32_t id : set.tree.primIndices)
                _instPrims.push_back(slotOf[K + set.firstInst + id]);
            // inner nodes only become BVH2 nodes; a leaf is a leaf reference in its parent
            std::vector<int32_t> nodeIndex(set.tree.nodes.size(), -1);
            int32_t next = int32_t(_nodes.size());
            for (size_t ni = 0; ni < set.tree.nodes.size(); ++ni)
                if (set.tree.nodes[ni].count == 0) nodeIndex[ni] = next++;
            auto refOf = [&](uint32_t ni) -> int32_t {
                const TgHipInstNode &n = set.tree.nodes[ni];
                return n.count == 0 ? nodeIndex[ni] : TGHIP_MAKE_LEAF(primBase + n.left, n.count);
            };
            for (size_t ni = 0; ni < set.tree.nodes.size(); ++ni) {
                const TgHipInstNode &n = set.tree.nodes[ni];
                if (n.count != 0) continue;
                TgHipBvhNode b;
                std::memset(&b, 0, sizeof(b));
                for (int k = 0; k < 3; ++k) {
                    b.lo0[k] = n.box[k*4 + 0]; b.lo1[k] = n.box[k*4 + 1];
                    b.hi0[k] = n.box[k*4 + 2]; b.hi1[k] = n.box[k*4 + 3];
                }
                b.child0 = refOf(n.left);
                b.child1 = refOf(n.left + 1);
                _nodes.push_back(b);
            }
            const int32_t root = refOf(0);
            std::memcpy(&_recs[K + N + si].c[0], &root, 4);
            refDepth = std::max(refDepth, set.tree.depth);
        }
    }
    const uint32_t numTopRecs = uint32_t(_recs.size());
    std::vector<uint32_t> masterRoot(masters.size(), 0), masterWideRoot(masters.size(), 0);
    int masterDepth = 0, masterWideDepth = 0;
    for (size_t mi = 0; mi < masters.size(); ++mi) {
        const std::vector<TgHipPrimRec> &mrecs = masters[mi].recs;
        const std::vector<TgHipTriAttr> &mattrs = masters[mi].attrs;
        const std::vector<Box3f> &mbounds = masters[mi].bounds;
        if (mrecs.empty()) continue;
        if (mattrs.size() != mrecs.size() || mbounds.size() != mrecs.size())
            throw std::runtime_error("master: one attribute record and one box per record");
        BvhBuildResult sub = buildBvh(mbounds, mbounds.size() < (1u << 18) ? 1 : 4);
        const uint32_t recBase = uint32_t(_recs.size()), nodeBase = uint32_t(_nodes.size());
        if (uint64_t(recBase) + mrecs.size() >= (1u << 27))
            throw std::runtime_error("too many primitive records for the 27-bit leaf encoding");
        // the master's wide subtree (only when the top level has one): built over the master's own records, then moved behind
        // the wide nodes and records that exist so far
        std::vector<uint32_t> place(sub.order);              // place[i] = the master's record that goes to slot recBase + i
        if (!_wideNodes.empty()) {
            std::vector<Box3f> ordered(mbounds.size());
            for (size_t i = 0; i < sub.order.size(); ++i)
                ordered[i] = mbounds[sub.order[i]];
            WideBvhResult wide = buildWideBvh(sub.nodes, ordered);
            if (wide.nodes.empty()) {
                _wideNodes.clear();                          // (a leaf too fat to collapse: the scene walks the BVH2)
                _wideDepth = 0;
            } else {
                for (size_t i = 0; i < wide.order.size(); ++i)
                    place[i] = sub.order[wide.order[i]];
                masterWideRoot[mi] = uint32_t(_wideNodes.size());
                for (TgHipWideNode w : wide.nodes) {
                    w.child_base += masterWideRoot[mi];
                    w.rec_base += recBase;
                    _wideNodes.push_back(w);
                }
                masterWideDepth = std::max(masterWideDepth, wide.depth);
            }
        }
        for (size_t i = 0; i < place.size(); ++i) {
            _recs.push_back(mrecs[place[i]]);
            _triAttrs.push_back(mattrs[place[i]]);
        }
        auto relocate = [&](int32_t ref) -> int32_t {
            if (ref >= 0) return ref + int32_t(nodeBase);
            return TGHIP_MAKE_LEAF(TGHIP_LEAF_FIRST(ref) + recBase, TGHIP_LEAF_COUNT(ref));
        };
        for (TgHipBvhNode n : sub.nodes) {
            n.child0 = relocate(n.child0);
            n.child1 = relocate(n.child1);
            _nodes.push_back(n);
        }
        masterRoot[mi] = nodeBase;
        masterDepth = std::max(masterDepth, sub.maxDepth);
    }
    {
        for (uint32_t i = 0; i < numTopRecs; ++i) {
            if (TGHIP_REC_KIND(_recs[i].meta) != TGHIP_REC_INSTANCE) continue;
            uint32_t slot;
            std::memcpy(&slot, &_recs[i].c[0], 4);
            std::memcpy(&_recs[i].c[0], &masterRoot[slot], 4);
            std::memcpy(&_recs[i].c[2], &masterWideRoot[slot], 4);   // root of the master's wide subtree (0: the scene has no wide BVH)
        }
        // the wide walk's stack: the groups of the top level, three entries where an instance is entered, the master's groups
        if (!_wideNodes.empty()) {
            _wideDepth += masterWideDepth + 3;
            if (_wideDepth > TGHIP_MAX_WIDE_DEPTH) { _wideNodes.clear(); _wideDepth = 0; }
        }
        // one device stack holds the scene's walk, above it the walk of the reference's instance tree and above that the walk of the
        // master being visited.  (BinaryBvh::trace keeps the distance at which the ray enters a stacked node and re-checks it when it
        // pops, bvh/BinaryBvh.hpp:277-283; the device recomputes it for a popped LEAF from inst_leaf_boxes and needs none for a popped
        // inner node, whose children all fail their own tests exactly when the node's check would: pt_kernels.h)
        _bvhDepth += refDepth + masterDepth + 3;
        if (_bvhDepth > TGHIP_MAX_BVH_DEPTH - 1)
            throw std::runtime_error("instanced BVH deeper than the device traversal stack");
    }
    out.numInstances = numInstances;
    out.numTopRecs = numTopRecs;
    return out;
}

void TraceableScene::flatten()
{
    auto t0 = std::chrono::steady_clock::now();

    // ---- prepareForRender pass (TraceableScene.hpp:64-102) -----------------------------------
    _scene.camera.precompute();
    for (auto &b : _scene.bsdfs)
        b->prepareForRender();
    _allPrims = _scene.primitives;
    int lightCount = 0;
    for (auto &p : _allPrims) {
        p->prepareForRender();
        for (auto &b : p->bsdfs)
            b->prepareForRender();
        if (p->isEmissive())
            lightCount++;
    }
    if (lightCount == 0) {
        // default white environment (TraceableScene.hpp:97-102)
        auto defaultLight = std::make_shared<Primitive>();
        defaultLight->type = Primitive::InfiniteSphere;
        defaultLight->name = "<default light>";
        defaultLight->emission = std::make_shared<Texture>();
        defaultLight->emission->value = Vec3f(1.0f);
        defaultLight->prepareForRender();
        _allPrims.push_back(defaultLight);
    }

    // ---- tables indexed by pointer -----------------------------------------------------------
    std::map<const Texture *, int32_t> texIndex;
    auto addTexture = [&](const std::shared_ptr<Texture> &t) -> int32_t {
        if (!t) return -1;
        auto it = texIndex.find(t.get());
        if (it != texIndex.end()) return it->second;
        TgHipTexture d;
        std::memset(&d, 0, sizeof(d));
        d.type = int32_t(t->type);
        copy3(d.value, t->value);
        copy3(d.on_color, t->onColor);
        copy3(d.off_color, t->offColor);
        d.res_u = t->resU; d.res_v = t->resV;
        d.scale = t->scale;
        copy3(d.avg, t->average());
        d.texel_offset = -1;
        d.dist_offset = -1;
        if (t->type == Texture::Bitmap) {
            d.w = t->w; d.h = t->h;
            d.flags = (t->linear ? TGHIP_TEXF_LINEAR : 0) | (t->clamp ? TGHIP_TEXF_CLAMP : 0) |
                      (t->rgb ? TGHIP_TEXF_RGB : 0) | (t->valid ? TGHIP_TEXF_VALID : 0);
            d.texel_offset = int64_t(_texels.size());
            _texels.insert(_texels.end(), t->texels.begin(), t->texels.end());
        }
        int32_t idx = int32_t(_textures.size());
        _textures.push_back(d);
        texIndex[t.get()] = idx;
        return idx;
    };
    auto addDistribution = [&](const std::shared_ptr<Texture> &t) {
        if (!t || t->type != Texture::Bitmap) return;
        t->makeSamplableSpherical();
        TgHipTexture &d = _textures[size_t(texIndex[t.get()])];
        if (d.dist_offset >= 0) return;
        d.dist_offset = int64_t(_dist.size());
        _dist.insert(_dist.end(), t->marginalPdf.begin(), t->marginalPdf.end());
        _dist.insert(_dist.end(), t->marginalCdf.begin(), t->marginalCdf.end());
        _dist.insert(_dist.end(), t->pdf.begin(), t->pdf.end());
        _dist.insert(_dist.end(), t->cdf.begin(), t->cdf.end());
    };

    std::map<const Bsdf *, int32_t> bsdfIndex;
    std::function<int32_t(const std::shared_ptr<Bsdf> &)> addBsdf = [&](const std::shared_ptr<Bsdf> &b) -> int32_t {
        if (!b) return -1;
        auto it = bsdfIndex.find(b.get());
        if (it != bsdfIndex.end()) return it->second;
        b->prepareForRender();
        int32_t idx = int32_t(_bsdfs.size());
        bsdfIndex[b.get()] = idx;
        _bsdfs.emplace_back();
        TgHipBsdf d;
        std::memset(&d, 0, sizeof(d));
        d.type = int32_t(b->type);
        d.lobes = b->lobes;
        d.albedo = addTexture(b->albedo);
        d.distribution = b->distribution;
        d.roughness = addTexture(b->roughness);
        d.sub0 = addBsdf(b->sub0);
        d.sub1 = addBsdf(b->sub1);
        d.tex1 = addTexture(b->tex1);
        d.bump1 = b->bump ? addTexture(b->bump) + 1 : 0;
        d.ior = b->ior; d.thickness = b->thickness;
        d.avg_transmittance = b->avgTransmittance;
        d.diffuse_fresnel = b->diffuseFresnel;
        d.enable_refraction = b->enableRefraction ? 1 : 0;
        copy3(d.eta, b->eta); copy3(d.k, b->k);
        copy3(d.sigma_a, b->sigmaA); copy3(d.scaled_sigma_a, b->scaledSigmaA);
        _bsdfs[size_t(idx)] = d;
        return idx;
    };
    for (auto &b : _scene.bsdfs)
        addBsdf(b);

    // ---- media (Scene::_media; inline media of primitives / the camera are added on first use) ----
    std::vector<const Medium *> mediumKeys;
    auto addMedium = [&](const std::shared_ptr<Medium> &m) -> int32_t {
        if (!m) return -1;
        for (size_t i = 0; i < mediumKeys.size(); ++i)
            if (mediumKeys[i] == m.get()) return int32_t(i);
        m->prepareForRender();
        TgHipMedium d;
        std::memset(&d, 0, sizeof(d));
        copy3(d.sigma_a, m->sigmaA); copy3(d.sigma_s, m->sigmaS); copy3(d.sigma_t, m->sigmaT);
        d.absorption_only = m->absorptionOnly ? 1 : 0;
        d.max_bounce = m->maxBounce;
        d.phase_type = m->phaseType;
        d.phase_g = m->phaseG;
        d.trans_type = m->transType;
        for (int k = 0; k < 3; ++k) d.trans_p[k] = m->transP[k];
        d.medium_type = m->mediumType;
        if (m->mediumType == TGHIP_MEDIUM_EXPONENTIAL) {
            d.falloff_scale = m->falloffScale;
            copy3(d.unit_point, m->unitPoint);
            copy3(d.falloff_dir, m->unitFalloffDirection);
        } else if (m->mediumType == TGHIP_MEDIUM_ATMOSPHERE) {
            // AtmosphericMedium::prepareForRender (AtmosphericMedium.cpp:66-84): the named pivot primitive's origin is the centre
            Vec3f center = m->center;
            if (!m->pivot.empty()) {
                const Primitive *pivot = nullptr;
                for (const auto &p : _scene.primitives)
                    if (p->name == m->pivot) { pivot = p.get(); break; }
                if (pivot)
                    center = pivot->transform*Vec3f(0.0f);
                else
                    std::fprintf(stderr, "Note: unable to find pivot object '%s' for atmospheric medium\n", m->pivot.c_str());
            }
            d.falloff_scale = m->effectiveFalloffScale;
            copy3(d.unit_point, center);
            d.falloff_dir[0] = m->radius;
        }
        mediumKeys.push_back(m.get());
        _media.push_back(d);
        const int32_t index = int32_t(_media.size() - 1);
        if (m->transType == TGHIP_TRANS_INTERPOLATED) {
            // the operands of an interpolated transmittance ride in the two entries behind it (include/tungsten_hip.h)
            for (int k = 0; k < 2; ++k) {
                TgHipMedium sub;
                std::memset(&sub, 0, sizeof(sub));
                sub.trans_type = m->subType[k];
                for (int j = 0; j < 3; ++j) sub.trans_p[j] = m->subP[k][j];
                mediumKeys.push_back(nullptr);
                _media.push_back(sub);
            }
        }
        return index;
    };
    for (auto &m : _scene.media)
        addMedium(m);

    // ---- objects, light lists, records -------------------------------------------------------
    std::vector<Box3f> recBounds;
    std::vector<std::shared_ptr<Primitive>> masterPrims;   // distinct master meshes of all `instances` primitives
    std::vector<InstanceSetInput> instSets;               // one per `instances` primitive that has instances of non-empty masters
    _sceneBounds = Box3f();
    for (size_t pi = 0; pi < _allPrims.size(); ++pi) {
        Primitive &p = *_allPrims[pi];
        TgHipObject o;
        std::memset(&o, 0, sizeof(o));
        o.type = p.type == Primitive::Skydome ? int32_t(TGHIP_OBJ_INFINITE_SPHERE) : int32_t(p.type);
        o.bsdf = p.bsdfs.empty() ? -1 : addBsdf(p.bsdfs[0]);
        for (size_t i = 1; i < p.bsdfs.size(); ++i) addBsdf(p.bsdfs[i]);
        bool emissive = p.isEmissive();
        o.emission = emissive ? addTexture(p.emission) : -1;
        o.light = -1;
        o.first_light_tri = -1;
        o.flags = (p.smooth ? TGHIP_OBJF_SMOOTH : 0) | (p.doSample ? TGHIP_OBJF_SAMPLE : 0) | (p.type == Primitive::Skydome ? TGHIP_OBJF_SKYDOME : 0);
        o.area = p.area; o.inv_area = p.invArea;
        copy3(o.base, p.base); copy3(o.edge0, p.edge0); copy3(o.edge1, p.edge1); copy3(o.normal, p.normal);
        o.inv_uv_sq[0] = p.invUvSq[0]; o.inv_uv_sq[1] = p.invUvSq[1];
        copy3(o.pos, p.pos); copy3(o.scale, p.scale);
        copyRot(o.rot, p.rot);
        copy3(o.face_cdf, p.faceCdf);
        o.int_medium = addMedium(p.intMedium);
        o.ext_medium = addMedium(p.extMedium);

        if (emissive) {
            if (p.isSamplable()) {
                if (p.type == Primitive::Mesh) {
                    // TriangleMesh::makeSamplable (TriangleMesh.cpp:395-409) + Distribution1D (sampling/Distribution1D.hpp:16-30)
                    const size_t n = p.tris.size();
                    if (n == 0)
                        throw std::runtime_error("emissive mesh '" + p.name + "' has no triangles");
                    std::vector<float> areas(n), cdf(n + 1);
                    float totalArea = 0.0f;
                    for (size_t i = 0; i < n; ++i) {
                        const MeshVertex &a = p.tfVerts[p.tris[i].v0], &b = p.tfVerts[p.tris[i].v1], &c = p.tfVerts[p.tris[i].v2];
                        Vec3f p0(a.pos[0], a.pos[1], a.pos[2]), p1(b.pos[0], b.pos[1], b.pos[2]), p2(c.pos[0], c.pos[1], c.pos[2]);
                        areas[i] = (p1 - p0).cross(p2 - p0).length()*0.5f;      // MathUtil::triangleArea
                        totalArea += areas[i];
                    }
                    cdf[0] = 0.0f;
                    for (size_t i = 0; i < n; ++i) cdf[i + 1] = cdf[i] + areas[i];
                    float totalWeight = cdf[n];
                    for (float &c : cdf) c /= totalWeight;
                    cdf[n] = 1.0f;
                    o.first_light_tri = int32_t(_lightTris.size());
                    o.num_light_tris = int32_t(n);
                    o.area = totalArea; o.inv_area = 1.0f/totalArea;
                    _lightTris.insert(_lightTris.end(), cdf.begin(), cdf.end());
                    for (size_t i = 0; i < n; ++i) {
                        const MeshVertex *v[3] = {&p.tfVerts[p.tris[i].v0], &p.tfVerts[p.tris[i].v1], &p.tfVerts[p.tris[i].v2]};
                        for (int k = 0; k < 3; ++k)
                            _lightTris.insert(_lightTris.end(), v[k]->pos, v[k]->pos + 3);
                    }
                }
                o.light = int32_t(_lights.size());
                _lights.push_back(int32_t(pi));
            }
            if (p.isInfinite())
                _infiniteLights.push_back(int32_t(pi));
        }
        _objects.push_back(o);

        if (p.isInfinite() || p.isDirac())
            continue;
        for (int k = 0; k < 3; ++k) _itemBoxes.push_back(p.bounds.lo[k]);
        for (int k = 0; k < 3; ++k) _itemBoxes.push_back(p.bounds.hi[k]);
        _itemObjects.push_back(int32_t(_objects.size() - 1));
        _sceneBounds.grow(p.bounds);

        uint32_t objMeta = uint32_t(pi);
        if (pi >= (1u << 29))
            throw std::runtime_error("too many primitives");
        switch (p.type) {
        case Primitive::Quad: {
            TgHipPrimRec r;
            std::memset(&r, 0, sizeof(r));
            copy3(r.a, p.base); copy3(r.b, p.edge0); copy3(r.c, p.edge1);
            r.p0 = p.invUvSq[0]; r.p1 = p.invUvSq[1];
            r.meta = (uint32_t(TGHIP_REC_QUAD) << 29) | objMeta;
            _recs.push_back(r);
            _triAttrs.emplace_back();
            std::memset(&_triAttrs.back(), 0, sizeof(TgHipTriAttr));
            _triAttrs.back().bsdf = o.bsdf;
            recBounds.push_back(p.bounds);
            break;
        } case Primitive::Cube: case Primitive::Sphere: case Primitive::Disk: case Primitive::Cylinder: {
            TgHipPrimRec r;
            std::memset(&r, 0, sizeof(r));
            copy3(r.a, p.pos); copy3(r.b, p.scale);
            r.meta = (uint32_t(p.type == Primitive::Cube ? TGHIP_REC_CUBE : p.type == Primitive::Sphere ? TGHIP_REC_SPHERE :
                               p.type == Primitive::Disk ? TGHIP_REC_DISK : TGHIP_REC_CYLINDER) << 29) | objMeta;
            _recs.push_back(r);
            _triAttrs.emplace_back();
            std::memset(&_triAttrs.back(), 0, sizeof(TgHipTriAttr));
            _triAttrs.back().bsdf = o.bsdf;
            recBounds.push_back(p.bounds);
            break;
        } case Primitive::Mesh: {
            std::vector<int32_t> meshBsdfs;
            for (auto &b : p.bsdfs) meshBsdfs.push_back(addBsdf(b));
            for (const MeshTriangle &t : p.tris) {
                const MeshVertex &a = p.tfVerts[t.v0], &b = p.tfVerts[t.v1], &c = p.tfVerts[t.v2];
                Vec3f p0(a.pos[0], a.pos[1], a.pos[2]), p1(b.pos[0], b.pos[1], b.pos[2]), p2(c.pos[0], c.pos[1], c.pos[2]);
                TgHipPrimRec r;
                std::memset(&r, 0, sizeof(r));
                copy3(r.a, p0); copy3(r.b, p1 - p0); copy3(r.c, p2 - p0);
                r.meta = (uint32_t(TGHIP_REC_TRIANGLE) << 29) | objMeta;
                _recs.push_back(r);
                TgHipTriAttr at;
                std::memcpy(at.n0, a.normal, 12); std::memcpy(at.n1, b.normal, 12); std::memcpy(at.n2, c.normal, 12);
                std::memcpy(at.uv0, a.uv, 8); std::memcpy(at.uv1, b.uv, 8); std::memcpy(at.uv2, c.uv, 8);
                at.bsdf = meshBsdfs[size_t(t.material)];
                _triAttrs.push_back(at);
                Box3f bb;
                bb.grow(p0); bb.grow(p1); bb.grow(p2);
                recBounds.push_back(bb);
            }
            break;
        } case Primitive::Instances: {
            // Instance::prepareForRender (Instance.cpp:392-428): one instance record per instance (position, rotation, its master's index
            // among `masterPrims`), the reference's box and the tight box of each; the trees are built by buildInstancedAccel below
            InstanceSetInput set;
            set.objMeta = objMeta;
            for (size_t i = 0; i < p.instancePos.size(); ++i) {
                const std::shared_ptr<Primitive> &m = p.masters[p.instanceId[i]];
                if (m->tris.empty() || m->verts.empty())
                    continue;                       // an empty master (e.g. an inline mesh, see Primitive::loadResources) is never hit
                size_t mi = 0;
                while (mi < masterPrims.size() && masterPrims[mi] != m) ++mi;
                if (mi == masterPrims.size())
                    masterPrims.push_back(m);
                TgHipPrimRec r;
                std::memset(&r, 0, sizeof(r));
                copy3(r.a, p.instancePos[i]);
                r.p0 = p.instanceRot[i][0];
                r.b[0] = p.instanceRot[i][1]; r.b[1] = p.instanceRot[i][2]; r.b[2] = p.instanceRot[i][3];
                uint32_t masterSlot = uint32_t(mi);
                std::memcpy(&r.c[0], &masterSlot, 4);
                r.meta = (uint32_t(TGHIP_REC_INSTANCE) << 29) | objMeta;
                set.recs.push_back(r);
                set.tightBounds.push_back(p.instanceBounds[i]);
                set.refBounds.push_back(p.instanceRefBounds[i]);
            }
            if (!set.recs.empty())
                instSets.push_back(std::move(set));
            break;
        } default:
            break;
        }
    }
    if (_recs.size() >= (1u << 27))
        throw std::runtime_error("too many primitive records for the 27-bit leaf encoding");

    // lights that are sampled need their 2-D distribution (TraceBase ctor -> makeSamplable,
    // integrators/TraceBase.cpp:5-22; InfiniteSphere.cpp:124-129)
    for (int32_t li : _lights)
        if (_allPrims[size_t(li)]->type == Primitive::InfiniteSphere || _allPrims[size_t(li)]->type == Primitive::Skydome)
            addDistribution(_allPrims[size_t(li)]->emission);           // (Skydome::makeSamplable, Skydome.cpp:138-143)

    // ---- BVH2 + the 8-wide BVH the single-level traversal kernels walk ----------------------------
    if (instSets.empty()) {
        SceneAccel accel = buildSceneAccel(_recs, _triAttrs, recBounds, false);
        _nodes.swap(accel.nodes);
        _wideNodes.swap(accel.wideNodes);
        _bvhDepth = accel.bvhDepth;
        _wideDepth = accel.wideDepth;
        _bvhSah = accel.sahCost;
        _desc.num_instances = 0;
        _desc.num_top_recs = uint32_t(_recs.size());
    } else {
        // ---- masters of instanced geometry: records in master space (the reference keeps an Embree scene per master mesh and transforms
        // the ray into it, Instance.cpp:290-311) ----
        std::vector<MasterInput> masters(masterPrims.size());
        for (size_t mi = 0; mi < masterPrims.size(); ++mi) {
            Primitive &m = *masterPrims[mi];
            // the master's object record (smooth flag, first bsdf); it is not a scene object of its own unless the scene also lists it
            size_t objIndex = 0;
            while (objIndex < _allPrims.size() && _allPrims[objIndex] != masterPrims[mi]) ++objIndex;
            if (objIndex == _allPrims.size()) {
                TgHipObject o;
                std::memset(&o, 0, sizeof(o));
                o.type = TGHIP_OBJ_MESH;
                o.int_medium = o.ext_medium = -1;
                o.bsdf = m.bsdfs.empty() ? -1 : addBsdf(m.bsdfs[0]);
                o.emission = -1; o.light = -1; o.first_light_tri = -1;
                o.flags = m.smooth ? TGHIP_OBJF_SMOOTH : 0;
                o.area = m.area; o.inv_area = m.invArea;
                objIndex = _objects.size();
                _objects.push_back(o);
            }
            std::vector<int32_t> meshBsdfs;
            for (auto &b : m.bsdfs) meshBsdfs.push_back(addBsdf(b));
            MasterInput &mo = masters[mi];
            for (const MeshTriangle &t : m.tris) {
                const MeshVertex &a = m.tfVerts[t.v0], &b = m.tfVerts[t.v1], &c = m.tfVerts[t.v2];
                Vec3f p0(a.pos[0], a.pos[1], a.pos[2]), p1(b.pos[0], b.pos[1], b.pos[2]), p2(c.pos[0], c.pos[1], c.pos[2]);
                TgHipPrimRec r;
                std::memset(&r, 0, sizeof(r));
                copy3(r.a, p0); copy3(r.b, p1 - p0); copy3(r.c, p2 - p0);
                r.meta = (uint32_t(TGHIP_REC_TRIANGLE) << 29) | uint32_t(objIndex);
                mo.recs.push_back(r);
                TgHipTriAttr at;
                std::memcpy(at.n0, a.normal, 12); std::memcpy(at.n1, b.normal, 12); std::memcpy(at.n2, c.normal, 12);
                std::memcpy(at.uv0, a.uv, 8); std::memcpy(at.uv1, b.uv, 8); std::memcpy(at.uv2, c.uv, 8);
                at.bsdf = meshBsdfs[size_t(t.material)];
                mo.attrs.push_back(at);
                Box3f bb;
                bb.grow(p0); bb.grow(p1); bb.grow(p2);
                mo.bounds.push_back(bb);
            }
        }
        InstancedAccel accel = buildInstancedAccel(_recs, _triAttrs, recBounds, instSets, masters);
        _nodes.swap(accel.nodes);
        _wideNodes.swap(accel.wideNodes);
        _instPrims.swap(accel.instPrims);
        _instLeafBoxes.swap(accel.instLeafBoxes);
        _instTightBoxes.swap(accel.instTightBoxes);
        _bvhDepth = accel.bvhDepth;
        _wideDepth = accel.wideDepth;
        _bvhSah = accel.sahCost;
        _desc.num_instances = accel.numInstances;
        _desc.num_top_recs = accel.numTopRecs;
    }

    // ---- camera / settings -------------------------------------------------------------------
    const Camera &cam = _scene.camera;
    TgHipCamera &c = _desc.camera;
    copy3(c.pos, cam.pos);
    c.plane_dist = cam.planeDist;
    copyRot(c.xf, cam.transform);
    c.ratio = cam.ratio;
    c.pixel_size_x = cam.pixelSizeX;
    c.res_x = int32_t(cam.resX); c.res_y = int32_t(cam.resY);
    c.filter_type = cam.filterType;
    c.filter_width = cam.filterWidth;
    c.filter_bin_size = cam.filterBinSize;
    std::memcpy(c.filter_cdf, cam.filterCdf, sizeof(c.filter_cdf));
    c.type = cam.thinlens ? TGHIP_CAMERA_THINLENS : cam.equirectangular ? TGHIP_CAMERA_EQUIRECTANGULAR : cam.cubemapMode >= 0 ? TGHIP_CAMERA_CUBEMAP : TGHIP_CAMERA_PINHOLE;
    c.focus_dist = cam.focusDist;
    c.aperture_size = cam.apertureSize;
    c.cat_eye = cam.catEye;
    c.aperture_type = cam.blades > 0 ? TGHIP_APERTURE_BLADE : TGHIP_APERTURE_DISK;
    if (cam.apertureTex) {
        // ThinlensCamera::precompute: _aperture->makeSamplable(MAP_UNIFORM) (cameras/ThinlensCamera.cpp:27-35); the forward path tracer
        // only ever samples the aperture, so its Distribution2D is all the device gets
        Texture &t = *cam.apertureTex;
        t.makeSamplable(false);
        if (t.marginalCdf.empty() || !(t.marginalCdf.back() > 0.0f) || !std::isfinite(t.marginalCdf.back()))
            throw std::runtime_error("thin lens camera: the aperture bitmap has no non-black texel to sample");
        c.aperture_type = TGHIP_APERTURE_BITMAP;
        c.aperture_w = t.w; c.aperture_h = t.h;
        c.aperture_dist = uint32_t(_dist.size());
        _dist.insert(_dist.end(), t.marginalPdf.begin(), t.marginalPdf.end());
        _dist.insert(_dist.end(), t.marginalCdf.begin(), t.marginalCdf.end());
        _dist.insert(_dist.end(), t.pdf.begin(), t.pdf.end());
        _dist.insert(_dist.end(), t.cdf.begin(), t.cdf.end());
    }
    c.blade_count = cam.blades;
    c.blade_angle = cam.bladeAngle; c.blade_step = cam.bladeStep;
    c.blade_edge[0] = cam.bladeEdge[0]; c.blade_edge[1] = cam.bladeEdge[1];
    for (int i = 0; i < 12; ++i) c.inv_xf[i] = cam.invTransform[i];
    if (cam.equirectangular || cam.cubemapMode >= 0) {
        // EquirectangularCamera::prepareForRender (cameras/EquirectangularCamera.cpp:129-134): _rot = _transform.extractRotation(); Camera::_pixelSize.y
        c.plane_dist = 0.0f;
        for (int i = 0; i < 12; ++i) c.inv_xf[i] = 0.0f;
        copyRot(c.inv_xf, cam.transform.extractRotation());
        c.inv_xf[9] = 1.0f/cam.resY;
        if (cam.cubemapMode >= 0) c.blade_count = cam.cubemapMode;     // (CubemapCamera.cpp:217-232: the face tables follow from the mode)
    }
    c.medium = addMedium(cam.medium);

    const IntegratorSettings &is = _scene.integrator;
    _desc.settings.min_bounces = is.minBounces;
    _desc.settings.max_bounces = is.maxBounces;
    _desc.settings.enable_light_sampling = is.enableLightSampling ? 1 : 0;
    _desc.settings.enable_two_sided_shading = is.enableTwoSidedShading ? 1 : 0;
    _desc.settings.enable_consistency_checks = is.enableConsistencyChecks ? 1 : 0;
    _desc.settings.enable_volume_light_sampling = is.enableVolumeLightSampling ? 1 : 0;
    if (!_media.empty() && (!is.lowOrderScattering || !is.includeSurfaces))
        throw std::runtime_error("path_tracer_hip renders media with low_order_scattering and include_surfaces at their defaults (true) only");

    _desc.abi_version = TGHIP_ABI_VERSION;
    _desc.num_nodes = uint32_t(_nodes.size());
    _desc.num_recs = uint32_t(_recs.size());
    _desc.num_objects = uint32_t(_objects.size());
    _desc.num_lights = uint32_t(_lights.size());
    _desc.num_infinite_lights = uint32_t(_infiniteLights.size());
    _desc.num_bsdfs = uint32_t(_bsdfs.size());
    _desc.num_textures = uint32_t(_textures.size());
    _desc.nodes = _nodes.data();
    _desc.recs = _recs.data();
    _desc.tri_attrs = _triAttrs.data();
    _desc.objects = _objects.data();
    _desc.lights = _lights.data();
    _desc.infinite_lights = _infiniteLights.data();
    _desc.bsdfs = _bsdfs.data();
    _desc.textures = _textures.data();
    _desc.media = _media.empty() ? nullptr : _media.data();
    _desc.inst_prims = _instPrims.empty() ? nullptr : _instPrims.data();
    _desc.num_inst_prims = uint32_t(_instPrims.size());
    _desc.inst_leaf_boxes = _instLeafBoxes.empty() ? nullptr : _instLeafBoxes.data();
    _desc.inst_tight_boxes = _instTightBoxes.empty() ? nullptr : _instTightBoxes.data();
    _desc.wide_nodes = _wideNodes.empty() ? nullptr : _wideNodes.data();
    _desc.num_wide_nodes = uint32_t(_wideNodes.size());
    // the reference's top-level Embree tree, for flat lists of analytic primitives (EmbreeTopTree.hpp): the visiting order where faces coincide
    // (renderer.scene_bvh = false: the reference commits no Embree scene and asks its finite primitives one after the other, in scene order,
    // renderer/TraceableScene.hpp:175-181 -- the plain record list, no tree)
    _topNodes.clear();
    if (_instPrims.empty() && _scene.renderer.useSceneBvh)
        _topNodes = buildSceneTopTree(_objects.data(), uint32_t(_objects.size()), _recs.data(), uint32_t(_recs.size()));
    _desc.top_nodes = _topNodes.empty() ? nullptr : _topNodes.data();
    _desc.num_top_nodes = uint32_t(_topNodes.size());
    _desc.num_media = uint32_t(_media.size());
    _desc.texels = _texels.data(); _desc.num_texel_floats = _texels.size();
    _desc.dist = _dist.data();     _desc.num_dist_floats = _dist.size();
    _desc.light_tris = _lightTris.data(); _desc.num_light_tri_floats = _lightTris.size();
    if (_scene.renderer.useSobol) {
        // RendererSettings::useSobol (renderer/RendererSettings.hpp:172-175): tiles get SobolPathSamplers
        const std::vector<uint32_t> &matrices = SobolMatrices::get();
        _desc.sobol_matrices = matrices.data();
        _desc.num_sobol_words = matrices.size();
    }
    copy3(_desc.bounds_lo, _sceneBounds.lo);
    copy3(_desc.bounds_hi, _sceneBounds.hi);

    _buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace tungsten_amd
