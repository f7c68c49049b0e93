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
} TgHostSceneInfo;

/* Scene::load + loadResources + flatten (Scene.cpp:378-391,281-306; TraceableScene.hpp:57-137) */
tgh_scene *tgh_scene_load(const char *json_path, char *err, size_t errlen);
const TgHipSceneDesc *tgh_scene_desc(tgh_scene *s);
int  tgh_scene_info(tgh_scene *s, TgHostSceneInfo *out);
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
void tgh_renderer_close(tgh_renderer *r);

/* file-format helpers used by the tests */
int tgh_save_pfm(const char *path, const float *rgb, int w, int h);
int tgh_load_hdr(const char *path, float *rgb /* may be NULL to query size */, int *w, int *h);

#ifdef __cplusplus
}
#endif
#endif
