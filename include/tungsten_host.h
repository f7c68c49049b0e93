/*
 * tungsten_host.h -- C view of the C++11 host side (scene loading, flattening, the
 * Integrator render loop) so that tests/bench (Python ctypes) and foreign hosts can drive it.
 * Mirrors what the reference's CLI does around the plugin surface
 * (src/tungsten/Shared.hpp:191-337: Scene::load -> loadResources -> makeTraceable(seed) ->
 * while (!integrator.done()) { startRender; waitForCompletion; } -> saveOutputs).
 * All functions return 0 on success, negative on error with a message in `err`.
 */
#ifndef TUNGSTEN_HOST_H_
#define TUNGSTEN_HOST_H_

#include "tungsten_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgh_scene tgh_scene;        /* Scene + TraceableScene (flattened, BVH built), no device */
typedef struct tgh_renderer tgh_renderer;  /* the above + PathTraceHipIntegrator bound to HIP device(s) */

typedef struct TgHostSceneInfo {
    uint32_t width, height, spp, spp_step;
    uint32_t num_nodes, num_recs, num_objects, num_lights, num_bsdfs, num_textures;
    int32_t  bvh_depth;
    double   bvh_sah_cost, build_seconds;
    int32_t  adaptive_sampling, stratified_sampler;
    uint32_t current_spp;      /* tgh_renderer_info only: Integrator::currentSpp() */
} TgHostSceneInfo;

/* Scene::load + loadResources + flatten (Scene.cpp:378-391,281-306; TraceableScene.hpp:57-137) */
tgh_scene *tgh_scene_load(const char *json_path, char *err, size_t errlen);
const TgHipSceneDesc *tgh_scene_desc(tgh_scene *s);
int  tgh_scene_info(tgh_scene *s, TgHostSceneInfo *out);
/* The items of the reference's top-level Embree geometry (renderer/TraceableScene.hpp:101-118): the scene's finite primitives in scene order, each
 * with the box its bounds() returns after prepareForRender -- restated per primitive class in csrc/host/Scene.cpp, held to the reference's own by
 * tests/test_top_tree.py -- as 6 floats (lower, upper), and the index of its object.  Returns the number of items; the arrays live as long as `s`. */
uint32_t tgh_scene_items(tgh_scene *s, const float **boxes, const int32_t **objects);
void tgh_scene_free(tgh_scene *s);

/* makeTraceable(seed) with the path_tracer_hip integrator (needs a HIP device) */
tgh_renderer *tgh_renderer_open(const char *json_path, uint32_t seed, int spp_override, int devices,
                                char *err, size_t errlen);
tghip_ctx *tgh_renderer_context(tgh_renderer *r, int device);
int  tgh_renderer_info(tgh_renderer *r, TgHostSceneInfo *out);
/* one pass: startRender + waitForCompletion; *done_out = integrator.done() */
int  tgh_renderer_step(tgh_renderer *r, int *done_out, char *err, size_t errlen);
/* while (!done) step; returns wall seconds of the loop in *seconds (Shared.hpp:281-315) */
int  tgh_renderer_render(tgh_renderer *r, double *seconds, char *err, size_t errlen);
int  tgh_renderer_image(tgh_renderer *r, float *rgb_mean, float *rgb_sum, uint32_t *count, size_t npixels,
                        char *err, size_t errlen);
int  tgh_renderer_save_outputs(tgh_renderer *r, char *err, size_t errlen);
/* Integrator::saveRenderResumeData / resumeRender (integrators/Integrator.cpp:108-162) on renderer.resume_render_file:
 * current spp, sampler flags, a hash of the flattened scene, the framebuffer and the integrator state (SampleRecords,
 * the scheduler's sampler).  *resumed_out = 1 when a matching state was found and restored (call before the first step). */
int  tgh_renderer_save_resume_data(tgh_renderer *r, char *err, size_t errlen);
int  tgh_renderer_resume(tgh_renderer *r, int *resumed_out, char *err, size_t errlen);
void tgh_renderer_close(tgh_renderer *r);

/* ---- pass scheduling (PathTraceIntegrator.cpp:27-134) -------------------------------------------------
 * SampleRecord as the integrator holds it (path_tracer/SampleRecord.hpp:13-16).  sample_count, mean and
 * running_variance are accumulated on the device (TGHIP_PASS_RECORDS); the rest is host state. */
typedef struct TgHostSampleRecord {
    uint32_t sample_count, next_sample_count, sample_index;
    float    adaptive_weight, mean, running_variance;
} TgHostSampleRecord;

/* records of the renderer after the last pass: n = ceil(W/4)*ceil(H/4) */
int  tgh_renderer_records(tgh_renderer *r, TgHostSampleRecord *out, size_t n, char *err, size_t errlen);

/* the renderer's auxiliary output buffers (renderer.output_buffers; cameras/OutputBuffer.hpp), one TgHipAuxPixel per pixel,
 * merged over its devices; all zero when the scene requests none */
int  tgh_renderer_output_buffers(tgh_renderer *r, TgHipAuxPixel *out, size_t npixels, char *err, size_t errlen);

/* The scheduler on its own (no device): tile seeds + generateWork over caller-supplied record statistics. */
typedef struct tgh_scheduler tgh_scheduler;
tgh_scheduler *tgh_scheduler_create(uint32_t width, uint32_t height, uint32_t seed);
size_t tgh_scheduler_num_tiles(tgh_scheduler *s);
size_t tgh_scheduler_num_records(tgh_scheduler *s);
const uint32_t *tgh_scheduler_tile_seeds(tgh_scheduler *s);
TgHostSampleRecord *tgh_scheduler_records(tgh_scheduler *s);   /* mutable view */
/* returns 1 when the pass has work, 0 when not (PathTraceIntegrator.cpp:108-134) */
int  tgh_scheduler_generate_work(tgh_scheduler *s, uint32_t current_spp, uint32_t next_spp, int adaptive);
/* the state of the scheduler's own sampler -- the one distributeAdaptiveSamples draws from (PathTraceIntegrator.cpp:93-134) --: what a
 * render-resume state holds next to the records (Integrator::saveState / loadState of a host that drives the scheduler through this API) */
uint64_t tgh_scheduler_sampler_state(tgh_scheduler *s);
void tgh_scheduler_set_sampler_state(tgh_scheduler *s, uint64_t state);
void tgh_scheduler_free(tgh_scheduler *s);
/* the Sobol' generator matrices the host hands to the device (NULL + message when the data file is missing) */
const uint32_t *tgh_sobol_matrices(size_t *num_words, char *err, size_t errlen);

/* ---- the acceleration-structure half of the flattening, on its own ------------------------------------
 * For hosts that fill a TgHipSceneDesc from their OWN scene classes -- the reference-side binding of INTEGRATION.md
 * section 3 (oracle/ref_binding/HipSceneFlattener.cpp walks Tungsten's TraceableScene) -- and want exactly the trees this
 * library's own loader builds: the binned-SAH BVH2 over the records' boxes (tghip nodes) collapsed into the 8-wide BVH
 * (wide_nodes; none for flat-list scenes of <= TGHIP_FLAT_MAX_RECS records).  Replaces the rtcCommit of the reference's
 * TraceableScene / TriangleMesh (renderer/TraceableScene.hpp:112-134, primitives/TriangleMesh.cpp:565).
 * `recs` and `tri_attrs` (num_recs entries each, single-level scenes: no instance records) are PERMUTED IN PLACE into the
 * order both trees refer to; bounds = 6 floats per record (lo.xyz, hi.xyz) in the caller's original order. */
typedef struct tgh_accel tgh_accel;
tgh_accel *tgh_accel_build(TgHipPrimRec *recs, TgHipTriAttr *tri_attrs, const float *bounds, uint32_t num_recs, char *err, size_t errlen);
const TgHipBvhNode  *tgh_accel_nodes(tgh_accel *a, uint32_t *num_nodes);
const TgHipWideNode *tgh_accel_wide_nodes(tgh_accel *a, uint32_t *num_wide_nodes);   /* NULL / 0: walk the BVH2 */
void tgh_accel_free(tgh_accel *a);

/* The same for a scene with `instances` primitives (Instance.cpp): the trees of include/tungsten_hip.h's instance-set layout -- the scene's
 * BVH2 with the reference's own tree over the instances behind it (restated node for node: the walk's visiting order is the reference's),
 * the wide BVH over the non-instance and the instance records, every master's two subtrees -- from plain arrays.
 *   recs / tri_attrs / bounds (num_recs, may be 0): the scene's non-instance records;
 *   sets: one per `instances` primitive -- its TGHIP_REC_INSTANCE records (a = position, p0 | b = rotation quaternion w | x y z, c[0] = index
 *         into `masters` as uint32 bits, meta = kind | the primitive's object index), and per instance the reference's box (the master box's
 *         eight rotated corners, Instance.cpp:409-421) and the tight box of its geometry (tgh_instance_tight_bounds);
 *   masters: each master mesh's triangle records in master space (meta = kind | the master's object index), attributes and boxes.
 * The result's record array (tgh_accel_recs / tgh_accel_tri_attrs) is the scene's whole array in the trees' order. */
typedef struct {
    uint32_t object, num_instances;
    const TgHipPrimRec *recs;
    const float *ref_bounds;        /* 6 floats per instance: lo.xyz, hi.xyz */
    const float *tight_bounds;      /* 6 floats per instance */
} TghInstanceSet;
typedef struct {
    const TgHipPrimRec *recs;
    const TgHipTriAttr *tri_attrs;
    const float *bounds;            /* 6 floats per record */
    uint32_t num_recs;
} TghMaster;
tgh_accel *tgh_accel_build_instanced(const TgHipPrimRec *recs, const TgHipTriAttr *tri_attrs, const float *bounds, uint32_t num_recs,
                                     const TghInstanceSet *sets, uint32_t num_sets, const TghMaster *masters, uint32_t num_masters,
                                     char *err, size_t errlen);
const TgHipPrimRec *tgh_accel_recs(tgh_accel *a, uint32_t *num_recs);                 /* instanced builds only (else NULL / 0) */
const TgHipTriAttr *tgh_accel_tri_attrs(tgh_accel *a);
const uint32_t *tgh_accel_inst_prims(tgh_accel *a, uint32_t *num_inst_prims);          /* TgHipSceneDesc::inst_prims */
const float *tgh_accel_inst_leaf_boxes(tgh_accel *a);                                  /* 8 floats per inst_prims slot */
const float *tgh_accel_inst_tight_boxes(tgh_accel *a);                                 /* 8 floats per top-level record */
void tgh_accel_counts(tgh_accel *a, uint32_t *num_top_recs, uint32_t *num_instances);
/* master_verts: 3 floats every stride_floats, in master space; rot = w x y z; ref_bounds / out = lo.xyz, hi.xyz */
void tgh_instance_tight_bounds(const float *master_verts, uint32_t stride_floats, uint32_t num_verts, const float pos[3], const float rot[4],
                               const float ref_bounds[6], float out[6]);

/* The reference's top-level Embree tree (include/tungsten_hip.h: TgHipTopNode) over n items given by their boxes (6 floats each: lower xyz,
 * upper xyz), as Embree 2.11's BVH4 builder for user geometry builds it (kernels/bvh/bvh_builder_sah.cpp:811-815 -> builders/bvh_builder_sah.h:
 * 176-282; restated in csrc/host/EmbreeTopTree.cpp).  Writes at most `capacity` nodes and returns the number the tree has (n - 1 at most;
 * 0 for n < 2 and for boxes Embree would drop as invalid), or -1 when `capacity` is too small. */
int tgh_top_tree_build(const float *boxes, uint32_t n, TgHipTopNode *nodes, uint32_t capacity);
/* TgHipSceneDesc::top_nodes for a flattened scene: the tree over the objects that have a record (in object order: the reference's _finites),
 * leaves naming records, when the scene is a flat list of quads / cubes / spheres / disks / cylinders; 0 nodes otherwise.  Same return convention. */
int tgh_top_tree_for_scene(const TgHipObject *objects, uint32_t num_objects, const TgHipPrimRec *recs, uint32_t num_recs,
                           TgHipTopNode *nodes, uint32_t capacity);
/* The box the reference's bounds callback reports for a record's primitive (TraceableScene.hpp:116-118): Quad::bounds / Cube::bounds /
 * Sphere::bounds / Disk::bounds / Cylinder::bounds (primitives/Quad.cpp:281-289, Cube.cpp:333-344, Sphere.cpp:273-276, Disk.cpp:298-306,
 * Cylinder.cpp:272-279) restated from the flattened object.  1 and the box, 0 for
 * a record kind (TGHIP_REC_*) whose bounds are not restated. */
int tgh_leaf_bounds(const TgHipObject *object, uint32_t kind, float lo[3], float hi[3]);

/* file-format helpers used by the tests */
int tgh_save_pfm(const char *path, const float *rgb, int w, int h);
int tgh_load_hdr(const char *path, float *rgb /* may be NULL to query size */, int *w, int *h);

#ifdef __cplusplus
}
#endif
#endif
