// Wavefront path-tracing kernels for gfx950 (MI355X).  Replaces, per (pixel, sample):
//   PathTraceIntegrator::renderTile            (integrators/path_tracer/PathTraceIntegrator.cpp:136-156)
//   PathTracer::traceSample                    (integrators/path_tracer/PathTracer.cpp:14-149)
//   TraceBase::{handleSurface, estimateDirect, lightSample, bsdfSample, generalizedShadowRay,
//               handleInfiniteLights}          (integrators/TraceBase.cpp)
//   TraceableScene::intersect + Embree         (renderer/TraceableScene.hpp:170-192)
//
// Execution model (DESIGN.md "Kernels"): the machine is partitioned into G persistent workgroups ("lanes of
// the wavefront tracer", G = CUs x blocks_per_cu, the same G for every kernel of a pass).  Workgroup b owns
//   * a private range of path slots in HBM (SoA arrays, slot = b*slots_per_block + local),
//   * private queue segments for those slots (extension / per-class shading / shadow), whose lengths live in
//     the workgroup's BlockCtl record, and
//   * a private, finely interleaved share of the pass's work items (item = pixel x chunk of sample indices).
// Nothing is shared between workgroups, so there is not a single global atomic on the hot path: queue pushes and
// work-item fetches are wave-aggregated (ballot + prefix popcount) LDS atomics, and the counts are carried from
// one kernel to the next through BlockCtl.  (Same-address global atomics saturate at ~88/us on MI355X --
// MI355X_MICROARCH.md "dequeue" -- which is what bounded the first version of this tracer.)
// A slot that finishes its item flushes the item's radiance sum to partial[item] and takes the workgroup's next
// item, so the pool stays full until the pass drains however uneven the path lengths are; partial[] is reduced
// per pixel in fixed chunk order by k_resolve, which keeps the image bit-reproducible.
// One wavefront iteration is
//     k_trace_closest -> k_shade<simple> [-> k_shade<complex>] -> k_trace_shadow
//   k_trace_closest  BVH2 closest hit for the extension queue (per-lane node stack in LDS); bins each path by the
//                    shading class of the surface it hit into one queue per class ("sort by material").
//   k_shade<M>       handleSurface for one class, compiled for the BSDF type set M only; emits <= 2 shadow
//                    rays and the continuation ray; finished paths are finalised and regenerated in place.
//   k_trace_shadow   generalizedShadowRay for the queued shadow rays; finishes the paths that were waiting
//                    for their last shadow result.
#ifndef TGAMD_PT_KERNELS_H_
#define TGAMD_PT_KERNELS_H_

#include "pt_scene.h"

// ---- slot state ------------------------------------------------------------------------------
enum { ST_DONE = 0, ST_ACTIVE = 1, ST_TERMINATED = 2, ST_TERMINATED_BLACK = 3 };
#define FLAG_BOUNCE(f)      ((f) & 0xFFu)
#define FLAG_SPECULAR       0x100u
#define FLAG_STATE(f)       (((f) >> 16) & 7u)
#define FLAG_MAKE(bounce, spec, state) ((uint32_t)(bounce) | ((spec) ? FLAG_SPECULAR : 0u) | ((uint32_t)(state) << 16))

#define PT_NUM_CLASSES 2          // shading classes: 0 = diffuse/null/miss, 1 = everything else
#define PT_ITEM_GROUP  64u        // consecutive work items handed to one workgroup (a wave's worth of pixels)

struct BlockCtl {                 // one per persistent workgroup; only that workgroup touches it
    uint32_t n_ext;               // extension-queue segment length
    uint32_t n_shade[PT_NUM_CLASSES];
    uint32_t n_shadow;
    uint32_t item_cursor;         // workgroup-local linear index of the next work item
    uint32_t pad[3];
    unsigned long long samples, closest_rays, shadow_rays, shadow_slots;
};                                // 64 B

struct BlockStats {               // traversal statistics (count_traversal option), one per workgroup
    unsigned long long nodes_visited, prims_tested, nodes_visited_shadow, prims_tested_shadow;
};

struct PathState {
    float4 *ray_o;     // origin.xyz, tmin
    float4 *ray_d;     // dir.xyz, tmax
    float4 *hit;       // t, u, v, record index (int bits; -1 = miss)
    float4 *thr;       // throughput.rgb, flags (uint bits)
    float4 *emi;       // radiance of the sample in flight
    float4 *acc;       // sum of the finished samples of the slot's current work item, count (uint bits)
    uint2  *rng;       // PCG state
    uint2  *samp;      // current sample index, end of the item's sample range
    uint32_t *pixel;   // pixel index of the current item
    uint32_t *item;    // current work item
    float4 *sh_o;      // shadow origin.xyz, epsilon
    float4 *sh_d0, *sh_c0;   // light-sample shadow ray: dir.xyz, tmax | unoccluded contribution, endCap|bounce bits
    float4 *sh_d1, *sh_c1;   // bsdf-sample shadow ray
    float4 *sh_w;      // throughput at the NEE vertex, light-selection weight
    float4 *sh_p;      // emission picked up at the same vertex (added after the NEE term), path flags (uint bits)
    uint32_t *q_ext, *q_shadow;            // queue segments: workgroup b uses [b*slots_per_block, (b+1)*slots_per_block)
    uint32_t *q_shade[PT_NUM_CLASSES];
    float4 *partial;   // per work item: radiance sum, count (uint bits)
    BlockCtl *ctl;
    BlockStats *stats;
    uint32_t *live;    // [0] = tag of the last iteration that left work in some extension queue; [1] = abort flag
    uint32_t num_slots, slots_per_block;
};

struct PassParams {
    uint32_t spp_begin, spp_end, seed;
    uint32_t chunk;            // samples per work item
    uint32_t chunks;           // items per pixel slot = ceil((spp_end - spp_begin)/chunk)
    uint32_t pix_slots;        // pixel slots in this batch (= tiles in batch * 256)
    uint32_t total_items;      // pix_slots*chunks
    uint32_t first_tile;       // first owned tile of the batch (index into the shard's tile list)
    uint32_t shard_index, shard_count;
    uint32_t tiles_x, num_tiles;
    uint32_t width, height;
};

PT_DEV uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave-aggregated queue push: one (LDS) atomic per wave (ballot + prefix popcount); `queue` is the workgroup's segment
PT_DEV void queuePush(bool push, uint32_t value, uint32_t *queue, uint32_t *counter)
{
    unsigned long long mask = __ballot(push);
    if (mask == 0ull)
        return;
    uint32_t lane = laneId();
    uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    int leader = __ffsll((long long)mask) - 1;
    if ((int)lane == leader)
        base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (push)
        queue[base + prefix] = value;
}

// pixel slot j of the batch -> image pixel (16x16 tile dicing, PathTraceIntegrator.cpp:27-42; tiles of a shard
// are dealt round-robin).  False for slots of an edge tile that fall outside the image.
PT_DEV bool slotPixel(const PassParams &pp, uint32_t j, uint32_t &x, uint32_t &y)
{
    uint32_t tileLocal = j >> 8, inTile = j & 255u;
    uint32_t tile = pp.shard_index + (pp.first_tile + tileLocal)*pp.shard_count;
    if (tile >= pp.num_tiles)
        return false;
    uint32_t tx = tile % pp.tiles_x, ty = tile/pp.tiles_x;
    x = tx*16u + (inTile & 15u);
    y = ty*16u + (inTile >> 4);
    return x < pp.width && y < pp.height;
}

// PinholeCamera::sampleDirection + ReconstructionFilter::sample (PinholeCamera.cpp:70-86,
// ReconstructionFilter.hpp:86-103,152-169).  Consumes two random numbers.
PT_DEV float filterSample1D(const TgHipCamera &cam, float xi)
{
    bool negative = xi < 0.5f;
    xi = negative ? xi*2.0f : (xi - 0.5f)*2.0f;
    int idx = 30;
    for (int i = 0; i < 30; ++i) {
        if (xi < cam.filter_cdf[i]) { idx = i; break; }
    }
    float pdf = cam.filter_cdf[idx] - cam.filter_cdf[idx - 1];
    float u = cam.filter_bin_size*(idx + (xi - cam.filter_cdf[idx - 1])/pdf);
    return negative ? -u : u;
}
PT_DEV void cameraRay(const TgHipCamera &cam, uint32_t px, uint32_t py, Rng &rng, f3 &o, f3 &d)
{
    float xi0 = rngNext1D(rng), xi1 = rngNext1D(rng);
    float fu = 0.0f, fv = 0.0f;
    if (cam.filter_type == TGHIP_FILTER_BOX) { fu = xi0 - 0.5f; fv = xi1 - 0.5f; }
    else if (cam.filter_type == TGHIP_FILTER_TABULATED) { fu = filterSample1D(cam, xi0); fv = filterSample1D(cam, xi1); }
    f3 localD = normalized(mk3(-1.0f + ((float)px + 0.5f + fu)*2.0f*cam.pixel_size_x,
                               cam.ratio - ((float)py + 0.5f + fv)*2.0f*cam.pixel_size_x,
                               cam.plane_dist));
    o = ld3(cam.pos);
    d = mat3Mul(cam.xf, localD);
}

// ---- BVH2 traversal (closest hit) ----------------------------------------------------------
PT_DEV bool boxTest(f3 lo, f3 hi, const RayD &ray, f3 invD, float tmax, float &tEntry)
{
    float t0x = (lo.x - ray.o.x)*invD.x, t1x = (hi.x - ray.o.x)*invD.x;
    float t0y = (lo.y - ray.o.y)*invD.y, t1y = (hi.y - ray.o.y)*invD.y;
    float t0z = (lo.z - ray.o.z)*invD.z, t1z = (hi.z - ray.o.z)*invD.z;
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), ray.tmin));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    tf *= 1.0000004f;
    tEntry = tn;
    return tn <= tf;
}

// `stack` is this lane's column of the workgroup's LDS stack: stack[level*stride]
template<bool COUNT>
PT_DEV float4 traverseClosest(const DeviceScene &s, const RayD &ray, int *stack, int stride,
                              uint32_t &nodesVisited, uint32_t &primsTested)
{
    float tmax = ray.tmax;
    float4 hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
    f3 invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
    int sp = 0;
    int cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 *n = s.nodes + (size_t)cur*4;
            float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            if (COUNT) nodesVisited++;
            float e0, e1;
            bool h0 = boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ray, invD, tmax, e0);
            bool h1 = boxTest(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), ray, invD, tmax, e1);
            int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            if (h0 && h1) {
                if (e1 < e0) { stack[sp*stride] = c0; cur = c1; }
                else { stack[sp*stride] = c1; cur = c0; }
                sp++;
                continue;
            } else if (h0) { cur = c0; continue; }
            else if (h1) { cur = c1; continue; }
        } else {
            uint32_t first = TGHIP_LEAF_FIRST(cur), count = TGHIP_LEAF_COUNT(cur);
            for (uint32_t i = first; i < first + count; ++i) {
                if (COUNT) primsTested++;
                testRecord(s, i, ray, tmax, hit);
            }
        }
        if (sp == 0)
            break;
        sp--;
        cur = stack[sp*stride];
    }
    return hit;
}

// wave-reduced statistics add into a workgroup-local (LDS) counter
PT_DEV void waveAddStat(uint32_t *dst, uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if (laneId() == 0 && v)
        atomicAdd(dst, v);
}

#endif
