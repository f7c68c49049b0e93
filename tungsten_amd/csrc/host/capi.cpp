// extern "C" view of the host classes (include/tungsten_host.h).
#include "../../../include/tungsten_host.h"

#include "EmbreeTopTree.hpp"
#include "ImageIO.hpp"
#include "Integrator.hpp"
#include "Sampling.hpp"
#include "Scene.hpp"
#include "TraceableScene.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>

using namespace tungsten_amd;

struct tgh_scene
{
    std::unique_ptr<Scene> scene;
    std::unique_ptr<TraceableScene> flattened;
};

struct tgh_scheduler
{
    PassScheduler scheduler;
};

struct tgh_accel
{
    SceneAccel accel;
    // tgh_accel_build_instanced: the trees (nodes / wide nodes moved into `accel`) and the scene's whole record array
    InstancedAccel inst;
    std::vector<TgHipPrimRec> recs;
    std::vector<TgHipTriAttr> attrs;
};

struct tgh_renderer
{
    std::unique_ptr<Scene> scene;
    std::shared_ptr<Integrator> integrator;
    std::unique_ptr<TraceableScene> flattened;
};

static void setErr(char *err, size_t errlen, const std::string &msg)
{
    if (err && errlen) {
        std::snprintf(err, errlen, "%s", msg.c_str());
    }
}

static void fillInfo(Scene &scene, TraceableScene &ts, TgHostSceneInfo *out)
{
    std::memset(out, 0, sizeof(*out));
    out->width = scene.camera.resX; out->height = scene.camera.resY;
    out->spp = scene.renderer.spp; out->spp_step = scene.renderer.sppStep;
    const TgHipSceneDesc &d = ts.desc();
    out->num_nodes = d.num_nodes; out->num_recs = d.num_recs; out->num_objects = d.num_objects;
    out->num_lights = d.num_lights; out->num_bsdfs = d.num_bsdfs; out->num_textures = d.num_textures;
    out->bvh_depth = ts.bvhDepth(); out->bvh_sah_cost = ts.bvhSahCost(); out->build_seconds = ts.buildSeconds();
    out->adaptive_sampling = scene.renderer.useAdaptiveSampling ? 1 : 0;
    out->stratified_sampler = scene.renderer.useSobol ? 1 : 0;
}

extern "C" {

tgh_scene *tgh_scene_load(const char *json_path, char *err, size_t errlen)
{
    try {
        std::unique_ptr<tgh_scene> s(new tgh_scene());
        s->scene = Scene::load(json_path);
        s->flattened.reset(new TraceableScene(*s->scene, nullptr, 0));
        return s.release();
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return nullptr;
    }
}

const TgHipSceneDesc *tgh_scene_desc(tgh_scene *s) { return s ? &s->flattened->desc() : nullptr; }

int tgh_scene_info(tgh_scene *s, TgHostSceneInfo *out)
{
    if (!s || !out) return -1;
    fillInfo(*s->scene, *s->flattened, out);
    return 0;
}

uint32_t tgh_scene_items(tgh_scene *s, const float **boxes, const int32_t **objects)
{
    if (!s || !s->flattened) return 0;
    if (boxes) *boxes = s->flattened->itemBoxes().data();
    if (objects) *objects = s->flattened->itemObjects().data();
    return uint32_t(s->flattened->itemObjects().size());
}

void tgh_scene_free(tgh_scene *s) { delete s; }

tgh_renderer *tgh_renderer_open(const char *json_path, uint32_t seed, int spp_override, int devices,
                                char *err, size_t errlen)
{
    try {
        std::unique_ptr<tgh_renderer> r(new tgh_renderer());
        r->scene = Scene::load(json_path);
        if (spp_override > 0) {
            // CLI --spp override (Shared.hpp:238-241); a single pass unless the scene asks otherwise
            r->scene->renderer.spp = uint32_t(spp_override);
        }
        if (devices > 0)
            r->scene->integrator.devices = devices;
        r->integrator = IntegratorFactory::instantiate(r->scene->integrator.type);
        // Scene::fromJson parsed the "integrator" block (Scene.cpp:251); hand it to the plugin
        if (PathTraceHipIntegrator *hip = dynamic_cast<PathTraceHipIntegrator *>(r->integrator.get()))
            hip->setSettings(r->scene->integrator);
        r->flattened.reset(new TraceableScene(*r->scene, r->integrator.get(), seed));   // Scene::makeTraceable
        return r.release();
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return nullptr;
    }
}

tghip_ctx *tgh_renderer_context(tgh_renderer *r, int device)
{
    if (!r) return nullptr;
    PathTraceHipIntegrator *hip = dynamic_cast<PathTraceHipIntegrator *>(r->integrator.get());
    return hip ? hip->context(size_t(device)) : nullptr;
}

int tgh_renderer_info(tgh_renderer *r, TgHostSceneInfo *out)
{
    if (!r || !out) return -1;
    fillInfo(*r->scene, *r->flattened, out);
    out->current_spp = r->integrator->currentSpp();
    return 0;
}

int tgh_renderer_step(tgh_renderer *r, int *done_out, char *err, size_t errlen)
{
    if (!r) return -1;
    try {
        r->integrator->startRender([]() {});
        r->integrator->waitForCompletion();
        if (done_out) *done_out = r->integrator->done() ? 1 : 0;
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return -1;
    }
}

int tgh_renderer_render(tgh_renderer *r, double *seconds, char *err, size_t errlen)
{
    if (!r) return -1;
    try {
        auto t0 = std::chrono::steady_clock::now();
        while (!r->integrator->done()) {       // Shared.hpp:283-293
            r->integrator->startRender([]() {});
            r->integrator->waitForCompletion();
        }
        if (seconds)
            *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return -1;
    }
}

int tgh_renderer_image(tgh_renderer *r, float *rgb_mean, float *rgb_sum, uint32_t *count, size_t npixels,
                       char *err, size_t errlen)
{
    if (!r) return -1;
    try {
        PathTraceHipIntegrator *hip = dynamic_cast<PathTraceHipIntegrator *>(r->integrator.get());
        const std::vector<float> &img = r->integrator->linearImage();
        if (img.size() != npixels*3) { setErr(err, errlen, "pixel count mismatch"); return -1; }
        if (rgb_mean) std::memcpy(rgb_mean, img.data(), img.size()*sizeof(float));
        if (hip && rgb_sum) std::memcpy(rgb_sum, hip->sumBuffer().data(), npixels*3*sizeof(float));
        if (hip && count) std::memcpy(count, hip->countBuffer().data(), npixels*sizeof(uint32_t));
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return -1;
    }
}

int tgh_renderer_save_outputs(tgh_renderer *r, char *err, size_t errlen)
{
    if (!r) return -1;
    try {
        r->integrator->saveOutputs();
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return -1;
    }
}

void tgh_renderer_close(tgh_renderer *r)
{
    if (!r) return;
    r->flattened.reset();     // ~TraceableScene -> integrator.teardownAfterRender (TraceableScene.hpp:139-160)
    delete r;
}

int tgh_renderer_save_resume_data(tgh_renderer *r, char *err, size_t errlen)
{
    if (!r) return -1;
    try {
        r->integrator->saveRenderResumeData();
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return -1;
    }
}

int tgh_renderer_resume(tgh_renderer *r, int *resumed_out, char *err, size_t errlen)
{
    if (!r) return -1;
    try {
        bool ok = r->integrator->supportsResumeRender() && r->integrator->resumeRender();
        if (resumed_out) *resumed_out = ok ? 1 : 0;
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return -1;
    }
}

int tgh_renderer_records(tgh_renderer *r, TgHostSampleRecord *out, size_t n, char *err, size_t errlen)
{
    if (!r || !out) return -1;
    PathTraceHipIntegrator *hip = dynamic_cast<PathTraceHipIntegrator *>(r->integrator.get());
    if (!hip) { setErr(err, errlen, "not a path_tracer_hip renderer"); return -1; }
    const std::vector<TgHostSampleRecord> &rec = hip->scheduler().records();
    if (rec.size() != n) { setErr(err, errlen, "record count mismatch"); return -1; }
    std::memcpy(out, rec.data(), n*sizeof(TgHostSampleRecord));
    return 0;
}

int tgh_renderer_output_buffers(tgh_renderer *r, TgHipAuxPixel *out, size_t npixels, char *err, size_t errlen)
{
    if (!r || !out) return -1;
    try {
        std::vector<TgHipAuxPixel> aux;
        r->integrator->currentOutputBuffers(aux);
        if (aux.size() != npixels) { setErr(err, errlen, "pixel count mismatch"); return -1; }
        std::memcpy(out, aux.data(), npixels*sizeof(TgHipAuxPixel));
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return -1;
    }
}

tgh_scheduler *tgh_scheduler_create(uint32_t width, uint32_t height, uint32_t seed)
{
    tgh_scheduler *s = new tgh_scheduler();
    s->scheduler.reset(width, height, seed);
    return s;
}
size_t tgh_scheduler_num_tiles(tgh_scheduler *s) { return s ? s->scheduler.tileSeeds().size() : 0; }
size_t tgh_scheduler_num_records(tgh_scheduler *s) { return s ? s->scheduler.records().size() : 0; }
const uint32_t *tgh_scheduler_tile_seeds(tgh_scheduler *s) { return s ? s->scheduler.tileSeeds().data() : nullptr; }
TgHostSampleRecord *tgh_scheduler_records(tgh_scheduler *s) { return s ? s->scheduler.records().data() : nullptr; }
int tgh_scheduler_generate_work(tgh_scheduler *s, uint32_t current_spp, uint32_t next_spp, int adaptive)
{
    if (!s) return -1;
    return s->scheduler.generateWork(current_spp, next_spp, adaptive != 0) ? 1 : 0;
}
uint64_t tgh_scheduler_sampler_state(tgh_scheduler *s) { return s ? s->scheduler.sampler().state() : 0; }
void tgh_scheduler_set_sampler_state(tgh_scheduler *s, uint64_t state) { if (s) s->scheduler.sampler().setState(state); }
void tgh_scheduler_free(tgh_scheduler *s) { delete s; }

const uint32_t *tgh_sobol_matrices(size_t *num_words, char *err, size_t errlen)
{
    try {
        const std::vector<uint32_t> &t = SobolMatrices::get();
        if (num_words) *num_words = t.size();
        return t.data();
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        if (num_words) *num_words = 0;
        return nullptr;
    }
}

tgh_accel *tgh_accel_build(TgHipPrimRec *recs, TgHipTriAttr *tri_attrs, const float *bounds, uint32_t num_recs, char *err, size_t errlen)
{
    try {
        if (!recs || !tri_attrs || !bounds || num_recs == 0)
            throw std::runtime_error("tgh_accel_build: no records");
        std::vector<TgHipPrimRec> r(recs, recs + num_recs);
        std::vector<TgHipTriAttr> a(tri_attrs, tri_attrs + num_recs);
        std::vector<Box3f> b(num_recs);
        for (uint32_t i = 0; i < num_recs; ++i) {
            if (TGHIP_REC_KIND(r[i].meta) == TGHIP_REC_INSTANCE)
                throw std::runtime_error("tgh_accel_build: instance records need the two-level build of this library's own loader");
            b[i].lo = Vec3f(bounds[6*i + 0], bounds[6*i + 1], bounds[6*i + 2]);
            b[i].hi = Vec3f(bounds[6*i + 3], bounds[6*i + 4], bounds[6*i + 5]);
        }
        std::unique_ptr<tgh_accel> out(new tgh_accel());
        out->accel = buildSceneAccel(r, a, b, false);
        std::memcpy(recs, r.data(), size_t(num_recs)*sizeof(TgHipPrimRec));
        std::memcpy(tri_attrs, a.data(), size_t(num_recs)*sizeof(TgHipTriAttr));
        return out.release();
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return nullptr;
    }
}

const TgHipBvhNode *tgh_accel_nodes(tgh_accel *a, uint32_t *num_nodes)
{
    if (num_nodes) *num_nodes = a ? uint32_t(a->accel.nodes.size()) : 0u;
    return a ? a->accel.nodes.data() : nullptr;
}

const TgHipWideNode *tgh_accel_wide_nodes(tgh_accel *a, uint32_t *num_wide_nodes)
{
    if (num_wide_nodes) *num_wide_nodes = a ? uint32_t(a->accel.wideNodes.size()) : 0u;
    return (a && !a->accel.wideNodes.empty()) ? a->accel.wideNodes.data() : nullptr;
}

void tgh_accel_free(tgh_accel *a)
{
    delete a;
}

static std::vector<Box3f> boxesOf(const float *bounds, size_t n)
{
    std::vector<Box3f> b(n);
    for (size_t i = 0; i < n; ++i) {
        b[i].lo = Vec3f(bounds[6*i + 0], bounds[6*i + 1], bounds[6*i + 2]);
        b[i].hi = Vec3f(bounds[6*i + 3], bounds[6*i + 4], bounds[6*i + 5]);
    }
    return b;
}

tgh_accel *tgh_accel_build_instanced(const TgHipPrimRec *recs, const TgHipTriAttr *tri_attrs, const float *bounds, uint32_t num_recs,
                                     const TghInstanceSet *sets, uint32_t num_sets, const TghMaster *masters, uint32_t num_masters,
                                     char *err, size_t errlen)
{
    try {
        if (num_recs && (!recs || !tri_attrs || !bounds))
            throw std::runtime_error("tgh_accel_build_instanced: records without attributes or bounds");
        if (!sets || num_sets == 0 || !masters || num_masters == 0)
            throw std::runtime_error("tgh_accel_build_instanced: no instance sets or no masters");
        std::unique_ptr<tgh_accel> out(new tgh_accel());
        out->recs.assign(recs, recs + num_recs);
        out->attrs.assign(tri_attrs, tri_attrs + num_recs);
        std::vector<InstanceSetInput> in(num_sets);
        for (uint32_t i = 0; i < num_sets; ++i) {
            if (sets[i].num_instances && (!sets[i].recs || !sets[i].ref_bounds || !sets[i].tight_bounds))
                throw std::runtime_error("tgh_accel_build_instanced: an instance set without records or bounds");
            in[i].objMeta = sets[i].object;
            in[i].recs.assign(sets[i].recs, sets[i].recs + sets[i].num_instances);
            in[i].refBounds = boxesOf(sets[i].ref_bounds, sets[i].num_instances);
            in[i].tightBounds = boxesOf(sets[i].tight_bounds, sets[i].num_instances);
        }
        std::vector<MasterInput> ms(num_masters);
        for (uint32_t i = 0; i < num_masters; ++i) {
            if (masters[i].num_recs && (!masters[i].recs || !masters[i].tri_attrs || !masters[i].bounds))
                throw std::runtime_error("tgh_accel_build_instanced: a master without records, attributes or bounds");
            ms[i].recs.assign(masters[i].recs, masters[i].recs + masters[i].num_recs);
            ms[i].attrs.assign(masters[i].tri_attrs, masters[i].tri_attrs + masters[i].num_recs);
            ms[i].bounds = boxesOf(masters[i].bounds, masters[i].num_recs);
        }
        out->inst = buildInstancedAccel(out->recs, out->attrs, boxesOf(bounds, num_recs), in, ms);
        out->accel.nodes.swap(out->inst.nodes);
        out->accel.wideNodes.swap(out->inst.wideNodes);
        out->accel.bvhDepth = out->inst.bvhDepth;
        out->accel.wideDepth = out->inst.wideDepth;
        out->accel.sahCost = out->inst.sahCost;
        return out.release();
    } catch (const std::exception &e) {
        setErr(err, errlen, e.what());
        return nullptr;
    }
}

const TgHipPrimRec *tgh_accel_recs(tgh_accel *a, uint32_t *num_recs)
{
    if (num_recs) *num_recs = a ? uint32_t(a->recs.size()) : 0u;
    return (a && !a->recs.empty()) ? a->recs.data() : nullptr;
}

const TgHipTriAttr *tgh_accel_tri_attrs(tgh_accel *a) { return (a && !a->attrs.empty()) ? a->attrs.data() : nullptr; }

const uint32_t *tgh_accel_inst_prims(tgh_accel *a, uint32_t *num_inst_prims)
{
    if (num_inst_prims) *num_inst_prims = a ? uint32_t(a->inst.instPrims.size()) : 0u;
    return (a && !a->inst.instPrims.empty()) ? a->inst.instPrims.data() : nullptr;
}

const float *tgh_accel_inst_leaf_boxes(tgh_accel *a) { return (a && !a->inst.instLeafBoxes.empty()) ? a->inst.instLeafBoxes.data() : nullptr; }
const float *tgh_accel_inst_tight_boxes(tgh_accel *a) { return (a && !a->inst.instTightBoxes.empty()) ? a->inst.instTightBoxes.data() : nullptr; }

void tgh_accel_counts(tgh_accel *a, uint32_t *num_top_recs, uint32_t *num_instances)
{
    if (num_top_recs) *num_top_recs = a ? a->inst.numTopRecs : 0u;
    if (num_instances) *num_instances = a ? a->inst.numInstances : 0u;
}

void tgh_instance_tight_bounds(const float *master_verts, uint32_t stride_floats, uint32_t num_verts, const float pos[3], const float rot[4],
                               const float ref_bounds[6], float out[6])
{
    Box3f ref;
    ref.lo = Vec3f(ref_bounds[0], ref_bounds[1], ref_bounds[2]);
    ref.hi = Vec3f(ref_bounds[3], ref_bounds[4], ref_bounds[5]);
    QuaternionF q(rot[0], rot[1], rot[2], rot[3]);
    Box3f t = tightInstanceBox(master_verts, stride_floats, num_verts, q, Vec3f(pos[0], pos[1], pos[2]), ref);
    for (int k = 0; k < 3; ++k) { out[k] = t.lo[k]; out[3 + k] = t.hi[k]; }
}

int tgh_top_tree_build(const float *boxes, uint32_t n, TgHipTopNode *nodes, uint32_t capacity)
{
    if (!boxes && n) return -1;
    std::vector<TopBox> in(n);
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { in[i].lo[k] = boxes[6*i + k]; in[i].hi[k] = boxes[6*i + 3 + k]; }
    const std::vector<TgHipTopNode> tree = buildEmbreeTopTree(in);
    if (tree.size() > capacity || (!tree.empty() && !nodes)) return -1;
    if (!tree.empty()) std::memcpy(nodes, tree.data(), tree.size()*sizeof(TgHipTopNode));
    return int(tree.size());
}

int tgh_top_tree_for_scene(const TgHipObject *objects, uint32_t num_objects, const TgHipPrimRec *recs, uint32_t num_recs,
                           TgHipTopNode *nodes, uint32_t capacity)
{
    if ((!objects && num_objects) || (!recs && num_recs)) return -1;
    const std::vector<TgHipTopNode> tree = buildSceneTopTree(objects, num_objects, recs, num_recs);
    if (tree.size() > capacity || (!tree.empty() && !nodes)) return -1;
    if (!tree.empty()) std::memcpy(nodes, tree.data(), tree.size()*sizeof(TgHipTopNode));
    return int(tree.size());
}

int tgh_leaf_bounds(const TgHipObject *object, uint32_t kind, float lo[3], float hi[3])
{
    return (object && lo && hi && referenceLeafBounds(*object, kind, lo, hi)) ? 1 : 0;
}

int tgh_save_pfm(const char *path, const float *rgb, int w, int h)
{
    return ImageIO::savePfm(path, rgb, w, h, 3) ? 0 : -1;
}

int tgh_load_hdr(const char *path, float *rgb, int *w, int *h)
{
    std::vector<float> data;
    std::string err;
    int iw = 0, ih = 0;
    if (!ImageIO::loadHdr(path, data, iw, ih, err)) return -1;
    if (w) *w = iw;
    if (h) *h = ih;
    if (rgb) std::memcpy(rgb, data.data(), data.size()*sizeof(float));
    return 0;
}

} // extern "C"
