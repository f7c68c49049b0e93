// path_tracer_hip: kernels + the extern "C" shim declared in include/tungsten_hip.h.
// gfx950 (MI355X) only.  See pt_kernels.h for the execution model and DESIGN.md for the layout.
#include "pt_kernels.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

// =============================================================================================
// Kernels
// =============================================================================================

// Section timers of the shading kernel (development aid; compiled in only with -DPT_PROFILE): s_memtime per
// wave at section boundaries, summed per workgroup into BlockStats::prof and printed by tghip_destroy.
#ifdef PT_PROFILE
#define PROF_DECL unsigned long long profT = clock64(), profAcc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(n) do { unsigned long long t_ = clock64(); profAcc[n] += t_ - profT; profT = t_; } while (0)
#define PROF_FLUSH(stats) do { if (laneId() == 0) for (int k_ = 0; k_ < 12; ++k_) atomicAdd(&(stats).prof[k_], profAcc[k_]); } while (0)
#else
#define PROF_DECL
#define PROF(n)
#define PROF_FLUSH(stats)
#endif

// BSDF type sets of the shading-kernel variants (pt_scene.h BsdfOps<D, M>)
#define TYPES_SIMPLE (BSDF_BIT(TGHIP_BSDF_LAMBERT) | BSDF_BIT(TGHIP_BSDF_NULL) | BSDF_BIT(TGHIP_BSDF_ERROR))
#define MASK_SIMPLE  (TYPES_SIMPLE | FEAT_ALL)
#define MASK_LEAN    TYPES_SIMPLE        /* analytic primitives, constant/checker textures, one area light (Cornell box) */
#ifndef SIMPLE_WAVES
#define SIMPLE_WAVES 3   /* measured: 4 waves/SIMD (128 VGPRs, spills) is 10 % slower on materialtest's k_shade */
#endif
#ifndef COAT_WAVES
#define COAT_WAVES   2   /* measured: 3 waves/SIMD is neutral on materialtest (480 vs 476 us), +3 % on mesh1m's k_shade */
#endif
#ifndef LEAN_WAVES
#define LEAN_WAVES   2   /* measured: 2 waves/SIMD without scratch beats 3 with 108 B of scratch (kernel is VALU-bound) */
#endif
#define MASK_COAT    (MASK_SIMPLE | BSDF_BIT(TGHIP_BSDF_ROUGH_CONDUCTOR) | BSDF_BIT(TGHIP_BSDF_SMOOTH_COAT) | \
                      BSDF_BIT(TGHIP_BSDF_MIRROR) | BSDF_BIT(TGHIP_BSDF_CONDUCTOR))
#define MASK_GLASS   (MASK_SIMPLE | BSDF_BIT(TGHIP_BSDF_DIELECTRIC) | BSDF_BIT(TGHIP_BSDF_ROUGH_DIELECTRIC) | \
                      BSDF_BIT(TGHIP_BSDF_MIRROR))

// Finalises the finished sample of every lane with `finished` set (OutputBuffer::addSample semantics,
// cameras/OutputBuffer.hpp:104-107: NaN/Inf samples are dropped without counting; PathTracer.cpp:119-122,
// 130-131: NaN radiance turns the sample black), moves the slot to its next sample or -- when its work item is
// exhausted -- flushes the item's sum and takes the workgroup's next item (`cursor` = the workgroup's LDS item
// cursor), and generates the next camera path in place.  `fresh` lanes own nothing yet (pass start).  Must be
// called by all lanes of the wave.  Returns true for lanes that now hold a new active path.
// SobolPathSampler::startPath (sampling/SobolPathSampler.hpp:47-52) for a path that starts or resumes at
// dimension `dim`: the tile's sampler seed (PathTraceIntegrator.cpp:27-42) scrambled by the pixel.
PT_DEV void rngStartSobol(Rng &rng, const DeviceScene &s, const PassParams &pp, uint32_t px, uint32_t py, uint32_t pixel,
                          uint32_t sample, uint32_t dim)
{
    uint32_t tile = (px >> 4) + (py >> 4)*pp.tiles_x;
    rng.sobol = s.sobol;
    rng.scramble = at32(pp.tile_seeds, tile) ^ hash32(pixel);
    rng.index = sample;
    rng.dim = dim;
}

// EXT: the pass may carry TGHIP_PASS_SOBOL / TGHIP_PASS_RECORDS state (checked at run time through pp.flags /
// pp.rec_count); false compiles those paths out (the specialised shading variants, DESIGN.md "Kernels").
template<bool CONVERGED = true, bool EXT = true>
PT_DEV bool nextPath(const DeviceScene &s, const PathState &st, const PassParams &pp, bool finished, bool fresh,
                     uint32_t slot, f3 em, bool black, uint32_t *cursor, bool aborted, uint32_t &finishedCount)
{
    uint2 samp = make_uint2(0u, 0u);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
    uint32_t pixel = 0, item = 0;
    uint32_t lumBase = 0;                  // EXT: index of sample `s` of this work item in pp.lum is lumBase + s
    const bool records = EXT && pp.rec_count != nullptr;
    bool want = fresh;
    if (finished) {
        uint4 sm = slotU4(st, A_SAMP, slot), misc = slotU4(st, A_MISC, slot);
        samp = make_uint2(sm.x, sm.y);
        lumBase = EXT ? sm.w : 0u;
        acc = slotF4(st, A_ACC, slot);
        pixel = misc.z;
        item = misc.w;
        if (black || isnan(sum3(em)))
            em = splat3(0.0f);
        if (records)   // SampleRecord::addSample(c) input (SampleRecord.hpp:55-58; Vec3f::luminance, math/Vec.hpp:195-199)
            at32(pp.lum, lumBase + samp.x) = em.x*0.2126f + em.y*0.7152f + em.z*0.0722f;
        if (!(isinf(em.x) || isinf(em.y) || isinf(em.z))) {
            acc.x += em.x; acc.y += em.y; acc.z += em.z;
            acc.w = __uint_as_float(__float_as_uint(acc.w) + 1u);
        }
        if constexpr (EXT) {
            if (pp.flags & TGHIP_PASS_AUX) {
                // the addSample calls of traceSample (PathTracer.cpp:78-96, 133-140), then the colour (PathTraceIntegrator.cpp:152)
                TgHipAuxPixel &px = pp.aux[pixel];
                float4 a0 = slotF4(st, A_AUX0, slot), a1 = slotF4(st, A_AUX1, slot);
                auxAdd3(px, TGHIP_AUX_DEPTH, 3, 1, a0.w, 0.0f, 0.0f);
                auxAdd3(px, TGHIP_AUX_NORMAL, 4, 3, a0.x, a0.y, a0.z);
                auxAdd3(px, TGHIP_AUX_ALBEDO, 7, 3, a1.x, a1.y, a1.z);
                auxAdd3(px, TGHIP_AUX_VISIBILITY, 10, 1, a1.w, 0.0f, 0.0f);
                auxAdd3(px, TGHIP_AUX_COLOR, 0, 3, em.x, em.y, em.z);
            }
        }
        finishedCount++;
        samp.x++;
        if (samp.x >= samp.y || aborted) {
            at32(st.partial, item) = acc;
            want = true;
        }
    }
    bool dead = false;
    for (;;) {
        uint32_t base = 0, rank = 0;
        if (CONVERGED) {
            // one LDS atomic per wave
            unsigned long long mask = __ballot(want);
            if (mask == 0ull)
                break;
            uint32_t lane = laneId();
            int leader = __ffsll((long long)mask) - 1;
            if ((int)lane == leader)
                base = atomicAdd(cursor, (uint32_t)__popcll(mask));
            base = __shfl(base, leader);
            rank = __popcll(mask & ((1ull << lane) - 1ull));
        } else {
            // called from divergent code (dynamic-fetch traversal): one LDS atomic per lane
            if (!want)
                break;
            base = atomicAdd(cursor, 1u);
        }
        if (want) {
            // workgroup-local index L -> item: groups of PT_ITEM_GROUP consecutive items are dealt round-robin
            uint32_t L = base + rank;
            uint64_t w64 = ((uint64_t)(L/PT_ITEM_GROUP)*gridDim.x + blockIdx.x)*PT_ITEM_GROUP + (L % PT_ITEM_GROUP);
            if (w64 >= pp.total_items || aborted) {
                want = false;
                dead = true;
            } else {
                uint32_t w = (uint32_t)w64;
                uint32_t c, j, x, y;
                bool inImage;
                uint32_t recOfItem = 0;
                if (records) {
                    // gap-free enumeration of a record pass (PassParams): chunk from the hint table, then the sorted pixel list
                    uint32_t wAbs = w + pp.item_base;
                    c = at32(pp.rec_hint, wAbs >> 6);
                    while (wAbs >= at32(pp.rec_chunk_start, c + 1u)) ++c;
                    j = wAbs - at32(pp.rec_chunk_start, c);
                    recOfItem = at32(pp.rec_sorted, j >> 4);
                    x = (recOfItem % pp.variance_w)*4u + (j & 3u);
                    y = (recOfItem/pp.variance_w)*4u + ((j >> 2) & 3u);
                    inImage = x < pp.width && y < pp.height;   // records on the right / bottom edge reach past the image
                } else {
                    c = w/pp.pix_slots; j = w - c*pp.pix_slots;
                    inImage = slotPixel(pp, j, x, y);
                }
                if (inImage) {
                    uint32_t rel = pp.spp_begin + c*pp.chunk, relEnd = pp.spp_end, first = 0u;
                    bool take = true;
                    if (records) {
                        // renderTile (PathTraceIntegrator.cpp:142-147): the pixel's record says which samples it traces
                        uint32_t cnt = at32(pp.rec_count, recOfItem);
                        first = at32(pp.rec_index, recOfItem);
                        rel = c*pp.chunk;
                        relEnd = cnt;
                        take = rel < relEnd;                     // (always, by construction of the enumeration)
                        lumBase = at32(pp.rec_lum, recOfItem) + (((y & 3u) << 2) | (x & 3u))*cnt - first;
                    }
                    if (take) {
                        want = false;
                        item = w;
                        pixel = x + y*pp.width;
                        samp.x = first + rel;
                        samp.y = first + min(rel + pp.chunk, relEnd);
                        acc = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
                    }
                }
            }
        }
    }
    bool push = false;
    if (finished || fresh) {
        if (dead) {
            slotF4(st, A_THR, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(FLAG_MAKE(0, 0, ST_DONE)));
        } else {
            Rng rng = rngStart(pp.seed, pixel, samp.x);          // PathSampleGenerator::startPath
            const uint32_t px = pixel % pp.width, py = pixel/pp.width;
            if (EXT && (pp.flags & TGHIP_PASS_SOBOL))
                rngStartSobol(rng, s, pp, px, py, pixel, samp.x, 0u);
            CameraRef cam = *asConst(s.camera);
            // thin-lens scenes run the EXT variants (the shim sets PT_PASS_THINLENS): the pinhole-only variants stay as lean as they were
            const bool lens = EXT && (pp.flags & PT_PASS_THINLENS) != 0u;
            float l0 = 0.0f, l1 = 0.0f;
            if (lens) { l0 = rngNext1DT<EXT>(rng); l1 = rngNext1DT<EXT>(rng); }   // the lens point is sampled first
            float xi0 = rngNext1DT<EXT>(rng), xi1 = rngNext1DT<EXT>(rng);
            f3 o, d;
            const bool cameraOk = cameraRay<EXT>(cam, lens, px, py, l0, l1, xi0, xi1, o, d);
            slotF4(st, A_RAY_O, slot) = mk4(o, 1e-4f);                      // Ray ctor default nearT (math/Ray.hpp:24)
            slotF4(st, A_RAY_D, slot) = mk4(d, cameraOk ? PT_INF : -1.0f);   // a failed camera sample: the ray can hit nothing ...
            slotU4(st, A_MISC, slot) = make_uint4((uint32_t)rng.state, (uint32_t)(rng.state >> 32), pixel, item);
            slotF4(st, A_EMI, slot) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            // ... and carries no throughput, so the escaped path adds nothing: a black sample (PathTracer.cpp:27-28)
            const float t0 = cameraOk ? 1.0f : 0.0f;
            uint32_t f0 = FLAG_MAKE(0, 1, ST_ACTIVE);                       // wasSpecular starts true
            if (EXT && (pp.flags & PT_PASS_MEDIA))
                f0 |= FLAG_MEDIUM_BITS(cam.medium, 0);                      // _scene->cam().medium(), state.reset() (PathTracer.cpp:38-41)
            slotF4(st, A_THR, slot) = make_float4(t0, t0, t0, __uint_as_float(f0));
            if (EXT && (pp.flags & TGHIP_PASS_AUX)) {                          // nothing recorded yet, hitDistance = 0
                const float nan = __uint_as_float(0x7FC00000u);
                slotF4(st, A_AUX0, slot) = make_float4(nan, nan, nan, 0.0f);
                slotF4(st, A_AUX1, slot) = make_float4(nan, nan, nan, nan);
            }
            slotU4(st, A_SAMP, slot) = make_uint4(samp.x, samp.y, EXT ? rng.dim : 0u, lumBase);   // .z: next Sobol' dimension
            slotF4(st, A_ACC, slot) = acc;
            push = true;
        }
    }
    return push;
}

// Pass start: every slot takes its first work item.
__global__ __launch_bounds__(256) void k_start(DeviceScene s, PathState st, PassParams pp)
{
    __shared__ BlockLds L;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) ctl.item_cursor = 0;
    __syncthreads();
    queuesBegin(L, st, ctl, -1, 0u, nullptr);
    const uint32_t first = blockIdx.x*st.slots_per_block;
    uint32_t finishedCount = 0;
    for (uint32_t base = 0; base < st.slots_per_block; base += blockDim.x) {
        uint32_t local = base + threadIdx.x;
        uint32_t slot = first + local;
        bool fresh = local < st.slots_per_block && slot < st.num_slots;
        bool push = nextPath(s, st, pp, false, fresh, slot, splat3(0.0f), false, &L.cursor, false, finishedCount);
        queuePush(push, local, L, Q_EXTP);
    }
    bool any = queuesEnd(L, st, -1, (1u << Q_COUNT) - 1u);   // every bitmap is (re)initialised here
    if (threadIdx.x == 0) {
        ctl.item_cursor = L.cursor;
        if (any) st.live[0] = 1u;
    }
}

// INST: the scene has instance records (two-level traversal; the instance of a hit goes to the spare word A_EMI.w)
// INST: 0 = single-level scene, 1 = instance records + every record kind, 2 = instance records in a scene of triangles and quads only
template<bool COUNT, bool FLAT, int INST = 0>
__global__ __launch_bounds__(512) void k_trace_closest(DeviceScene s, PathState st)
{
    extern __shared__ int ldsStack[];
    __shared__ BlockLds L;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    // the dynamic LDS region first holds the expanded queue, then (after orderPreload's barrier) the node stacks
    queuesBegin(L, st, ctl, Q_EXTP, 0u, reinterpret_cast<unsigned short *>(ldsStack), Q_EXT);   // the shading queues are empty here
    const uint32_t n = L.n;
    const OrderRegs ord = orderPreload(reinterpret_cast<unsigned short *>(ldsStack), n);
    const uint32_t first = blockIdx.x*st.slots_per_block;
    uint32_t nodes = 0, prims = 0, rays = 0;
    for (uint32_t base = 0, k = 0; base < n; base += blockDim.x, ++k) {
        uint32_t i = base + threadIdx.x;
        uint32_t slot = 0, local = 0;
        int cls = -1;
        if (i < n) {
            local = orderGet(ord, k);
            slot = first + local;
            float4 ro = slotF4(st, A_RAY_O, slot), rd = slotF4(st, A_RAY_D, slot);
            RayD ray;
            ray.o = xyz(ro); ray.d = xyz(rd); ray.tmin = ro.w; ray.tmax = rd.w;
            float4 hit;
            if (INST) {
                int hitInst;
                hit = traverseClosestInst<COUNT, INST == 2 ? KINDS_MESH : KINDS_ALL>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims, hitInst);
                slotF4(st, A_EMI, slot).w = __int_as_float(hitInst);
            } else {
                hit = traverseClosest<COUNT, FLAT>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims);
            }
            slotF4(st, A_HIT, slot) = hit;
            int ri = __float_as_int(hit.w);
            cls = ri < 0 ? 0 : (int)at32(s.rec_class, (uint32_t)ri);
            rays++;
        }
        // sort by material: one shading queue per class
        queuePush(cls == 0, local, L, Q_SHADE0);
        queuePush(cls == 1, local, L, Q_SHADE1);
    }
    waveAddStat(&L.closest_rays, rays);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }
    queuesEnd(L, st, Q_EXT, (1u << Q_SHADE0) | (1u << Q_SHADE1), Q_EXTP);
    if (threadIdx.x == 0) {
        ctl.closest_rays += L.closest_rays;
        if (COUNT) { st.stats[blockIdx.x].nodes_visited += L.nodes; st.stats[blockIdx.x].prims_tested += L.prims; }
    }
}

// BVH closest hit with dynamic ray fetch ("persistent threads" inside the workgroup): a lane whose ray has
// finished does not idle until the slowest lane of its wave is done -- once fewer than 3/4 of the wave's lanes
// are busy, the idle lanes take the next rays of the workgroup's queue (one wave-aggregated LDS atomic) and join
// the traversal loop.  One loop iteration advances every busy lane by one BVH node or one leaf.
// Dynamic LDS: [expanded queue, 2 B per slot][node stacks, bvhDepth ints per thread].
// SOLIDS: the scene has cube / sphere / disk records somewhere; without them only triangle and quad tests are compiled in
template<bool COUNT, bool SOLIDS = true>
__global__ __launch_bounds__(512) void k_trace_closest_dyn(DeviceScene s, PathState st)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    int *stack = ldsDyn + PT_MAX_SLOTS_PER_BLOCK/2 + threadIdx.x;
    const int stride = (int)blockDim.x;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) fetchNext = 0;
    queuesBegin(L, st, ctl, Q_EXTP, 0u, order, Q_EXT);
    const uint32_t n = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    uint32_t nodes = 0, prims = 0, rays = 0;

    bool busy = false;
    uint32_t slot = 0, local = 0;
    RayD ray; ray.o = splat3(0.0f); ray.d = splat3(1.0f); ray.tmin = 0.0f; ray.tmax = 0.0f;
    f3 invD = splat3(1.0f);
    float tmax = 0.0f;
    float4 hit = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    int cur = 0, sp = 0;
    bool exhausted = false;                      // wave-uniform: the queue has been handed out completely
    for (;;) {
        unsigned long long busyMask = __ballot(busy);
        if (!exhausted && __popcll(busyMask) <= 48) {
            // refill the idle lanes
            unsigned long long want = ~busyMask;
            uint32_t lane = laneId();
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&fetchNext, (uint32_t)__popcll(want));
            base = __shfl(base, 0);
            if (!busy) {
                uint32_t i = base + __popcll(want & ((1ull << lane) - 1ull));
                if (i < n) {
                    local = order[i];
                    slot = first + local;
                    float4 ro = slotF4(st, A_RAY_O, slot), rd = slotF4(st, A_RAY_D, slot);
                    ray.o = xyz(ro); ray.d = xyz(rd); ray.tmin = ro.w; ray.tmax = rd.w;
                    invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
                    tmax = ray.tmax;
                    hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
                    cur = 0; sp = 0;
                    busy = true;
                    rays++;
                }
            }
            if (base + (uint32_t)__popcll(want) >= n)
                exhausted = true;
            busyMask = __ballot(busy);
        }
        if (busyMask == 0ull)
            break;
        // "while-while": lanes take node steps until they reach a leaf, then wait; the (long) leaf code runs only when
        // enough lanes have one to process -- otherwise every iteration would pay for both the node and the leaf path
        // with a handful of active lanes each.
        bool pop = false;
        if (busy && cur >= 0) {
            const float4 *nd = &at32(s.nodes, (uint32_t)cur*4u);
            float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
            if (COUNT) nodes++;
            float e0, e1;
            bool h0 = boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ray, invD, tmax, e0);
            bool h1 = boxTest(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), ray, invD, tmax, e1);
            int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            if (h0 && h1) {
                if (e1 < e0) { stack[sp*stride] = c0; cur = c1; }
                else { stack[sp*stride] = c1; cur = c0; }
                sp++;
            } else if (h0) { cur = c0; }
            else if (h1) { cur = c1; }
            else pop = true;
        }
        {
            unsigned long long atLeaf = __ballot(busy && cur < 0 && !pop);
            unsigned long long atNode = __ballot(busy && (cur >= 0 || pop));
            // process leaves when a good part of the wave waits for it, or nobody has node work left
            if (atLeaf != 0ull && ((uint32_t)__popcll(atLeaf) >= st.leaf_batch || atNode == 0ull)) {
                if (busy && cur < 0 && !pop) {
                    uint32_t firstRec = TGHIP_LEAF_FIRST(cur), count = TGHIP_LEAF_COUNT(cur);
                    for (uint32_t r = firstRec; r < firstRec + count; ++r) {
                        if (COUNT) prims++;
                        testRecord<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, r, ray, tmax, hit);
                    }
                    pop = true;
                }
            }
        }
        if (busy && pop) {
            if (sp == 0) {
                // finished: publish the hit and bin the path by shading class
                slotF4(st, A_HIT, slot) = hit;
                int ri = __float_as_int(hit.w);
                int cls = ri < 0 ? 0 : (int)at32(s.rec_class, (uint32_t)ri);
                queuePush(true, local, L, cls == 0 ? Q_SHADE0 : Q_SHADE1);
                busy = false;
            } else {
                sp--;
                cur = stack[sp*stride];
            }
        }
    }
    waveAddStat(&L.closest_rays, rays);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }
    queuesEnd(L, st, Q_EXT, (1u << Q_SHADE0) | (1u << Q_SHADE1), Q_EXTP);
    if (threadIdx.x == 0) {
        ctl.closest_rays += L.closest_rays;
        if (COUNT) { st.stats[blockIdx.x].nodes_visited += L.nodes; st.stats[blockIdx.x].prims_tested += L.prims; }
    }
}

// stand-alone batched closest-hit query (tghip_trace_rays) on caller rays
template<bool COUNT, bool FLAT, int INST = 0>
__global__ __launch_bounds__(256) void k_trace_rays(DeviceScene s, const float4 *rays, float4 *hits, uint32_t n, BlockStats *stats)
{
    extern __shared__ int ldsStack[];
    __shared__ uint32_t ldsNodes, ldsPrims;
    if (threadIdx.x == 0) { ldsNodes = 0; ldsPrims = 0; }
    __syncthreads();
    const uint32_t stride = gridDim.x*blockDim.x;
    uint32_t nodes = 0, prims = 0;
    for (uint32_t i = blockIdx.x*blockDim.x + threadIdx.x; i < n; i += stride) {
        float4 ro = rays[i*2 + 0], rd = rays[i*2 + 1];
        RayD ray;
        ray.o = xyz(ro); ray.d = xyz(rd); ray.tmin = ro.w; ray.tmax = rd.w;
        int hitInst;      // TgHipHit reports the record that was hit, not the instance it was reached through
        hits[i] = INST ? traverseClosestInst<COUNT, INST == 2 ? KINDS_MESH : KINDS_ALL>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims, hitInst)
                       : traverseClosest<COUNT, FLAT>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims);
    }
    if (COUNT) {
        waveAddStat(&ldsNodes, nodes);
        waveAddStat(&ldsPrims, prims);
        __syncthreads();
        if (threadIdx.x == 0) { stats[blockIdx.x].nodes_visited += ldsNodes; stats[blockIdx.x].prims_tested += ldsPrims; }
    }
}

// PathTracer::traceSample's loop body for one vertex: TraceBase::handleSurface (TraceBase.cpp:516-568)
// with estimateDirect split into "compute the unoccluded contribution here, test visibility in
// k_trace_shadow", plus the loop epilogue (PathTracer.cpp:108-129).  M = BSDF types this variant handles.
// W = waves per SIMD the register allocator must leave room for (occupancy vs spilling: 3 costs the
// Lambert-only variant 12 B of scratch, the others stay at 2)
// FUSE (flat-list scenes without forward-lobe BSDFs only -- the whole scene is a handful of records read through the
// scalar cache, so a separate traversal launch would spend its time on path-state traffic):
//   FUSE_TRACE   the kernel consumes the extension queues itself, intersects the ray inline, shades class-0 hits
//                and forwards class-1 hits (hit record stored) to the class-1 shading queue;
//   FUSE_SHADOW  the <= 2 shadow rays of a vertex are any-hit tested inline instead of being queued for
//                k_trace_shadow, so the NEE term is added on the spot and no shadow record is written.
//   FUSE_LOOP    (with both of the above, scenes without class-1 materials) the workgroup runs its slots to completion
//                inside ONE launch: nothing it reads or writes is shared with another workgroup, so the wavefront
//                iterations need no grid-wide synchronisation -- the queues simply stay in LDS between iterations.
// record kinds a shading variant's fused traversal has to test: the lean variant's scenes hold quads and cubes only
constexpr uint32_t shadeKinds(uint32_t M)
{
    return (M & FEAT_SOLIDS) ? ((M & FEAT_CYLINDER) ? KINDS_ALL : (KINDS_ALL & ~KIND_BIT(TGHIP_REC_CYLINDER)))
                             : (KIND_BIT(TGHIP_REC_QUAD) | KIND_BIT(TGHIP_REC_CUBE) | ((M & FEAT_TRIANGLES) ? KIND_BIT(TGHIP_REC_TRIANGLE) : 0u));
}
// TGHIP_PASS_AUX: output values of a sample that leaves the loop of traceSample without having recorded any (PathTracer.cpp:133-140).
// `asked`: handleInfiniteLights ran for direction `dir` (so info.primitive is the infinite light it found, if any).
template<uint32_t M>
PT_DEV void auxPostLoop(const DeviceScene &s, f3 dir, bool asked, int bounce, float4 &aux0, float4 &aux1)
{
    aux0 = mk4(-dir, bounce == 0 ? 0.0f : __uint_as_float(0x7FC00000u));
    if (asked && (M & FEAT_INFINITE)) {
        int objIdx = -1;
        for (uint32_t li = 0; li < s.num_infinite_lights; ++li) {
            const TgHipObject &c = s.objects[s.infinite_lights[li]];
            if (c.type != TGHIP_OBJ_INFINITE_SPHERE_CAP || dot(dir, ld3(c.normal)) >= c.scale[0])
                objIdx = s.infinite_lights[li];
        }
        if (objIdx >= 0) {                       // info.primitive->isInfinite(): + evalDirect
            const TgHipObject &o = s.objects[objIdx];
            float u = 0.0f, v = 0.0f, sinTheta;
            if (o.type == TGHIP_OBJ_INFINITE_SPHERE) infDirectionToUV(o, dir, u, v, sinTheta);
            f3 e = textureEval<M>(s, o.emission, u, v);
            aux1.x = e.x; aux1.y = e.y; aux1.z = e.z;
        }
    }
}

#define FUSE_TRACE  1
#define FUSE_SHADOW 2
#define FUSE_LOOP   4
template<uint32_t M, int W, int FUSE>
__global__ __launch_bounds__(256, W) void k_shade(DeviceScene sg, PathState st, PassParams pp, int cls)
{
    __shared__ BlockLds L;
    __shared__ __attribute__((aligned(16))) unsigned char ldsTables[PT_LDS_TABLE_BYTES];
    __shared__ unsigned short order[PT_MAX_SLOTS_PER_BLOCK];
    BlockCtl &ctl = st.ctl[blockIdx.x];
    const int qIn = (FUSE & FUSE_TRACE) ? Q_EXTP : cls == 0 ? Q_SHADE0 : Q_SHADE1;
    const int qIn2 = (FUSE & FUSE_TRACE) ? Q_EXT : -1;
    const uint32_t appendMask = (1u << Q_EXT) | (1u << Q_EXTP) | (1u << Q_SHADOW) | ((FUSE & FUSE_TRACE) ? (1u << Q_SHADE1) : 0u);
    queuesBegin(L, st, ctl, qIn, appendMask, order, qIn2);
    const DeviceScene s = stageSceneTables(sg, ldsTables);
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const int maxBounces = s.settings.max_bounces, minBounces = s.settings.min_bounces;
    const bool nee = s.settings.enable_light_sampling != 0;
    uint32_t finishedCount = 0, fusedClosest = 0, fusedShadow = 0, fusedPrims = 0, fusedNodes = 0;
    PROF_DECL;

    // FUSE_LOOP runs without queues: a finished path is regenerated in place, so a slot stays busy until the workgroup's
    // work items run out.  Thread t owns slots t, t + blockDim, ... for the whole launch (consecutive lanes = consecutive
    // slots: every state access is a full cache line), `idle` has one bit per owned slot that has nothing left to do,
    // and every wave leaves the loop on its own -- no barrier, no bitmap traffic between the wavefront iterations.
    constexpr bool DIRECT = (FUSE & FUSE_LOOP) != 0;
    uint32_t idle = 0;
    if (DIRECT) {
        // queuesBegin expanded (and thereby cleared) the extension queues into order[0, L.n): turn that list back into
        // a bitmap of busy slots (in the unused Q_SHADE0 words) each thread can look its own slots up in
        for (uint32_t i = threadIdx.x; i < L.n; i += blockDim.x)
            queuePush(true, order[i], L, Q_SHADE0);
        __syncthreads();
        for (uint32_t k = 0, local = threadIdx.x; k < 32u; ++k, local += blockDim.x) {
            bool queued = local < st.slots_per_block && ((L.bm[Q_SHADE0][local >> 5] >> (local & 31u)) & 1u);
            idle |= queued ? 0u : (1u << k);
        }
    }

  for (;;) {                                     // one wavefront iteration per turn (a single turn unless FUSE_LOOP)
    const uint32_t n = DIRECT ? st.slots_per_block : L.n;
    const bool aborted = __hip_atomic_load(&st.live[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (uint32_t base = 0, turn = 0; base < n; base += blockDim.x, ++turn) {
        uint32_t i = base + threadIdx.x;
        PROF(0);
        bool hasShadow = false, finished = false, survives = false, black = false, toComplex = false;
        uint32_t slot = 0, local = 0;
        f3 em = splat3(0.0f);
        if (DIRECT ? !((idle >> turn) & 1u) : i < n) {
            f3 pendingOut = splat3(0.0f);
            local = DIRECT ? i : order[i];
            slot = first + local;
            float4 ro = slotF4(st, A_RAY_O, slot), rd = slotF4(st, A_RAY_D, slot), hit, thr4 = slotF4(st, A_THR, slot);
            if (FUSE & FUSE_TRACE) {
                // TraceableScene::intersect inline: the flat record list, walked uniformly by the wave
                RayD r0;
                r0.o = xyz(ro); r0.d = xyz(rd); r0.tmin = ro.w; r0.tmax = rd.w;
                hit = traverseClosest<true, true, shadeKinds(M)>(sg, r0, nullptr, 0, fusedNodes, fusedPrims);
                fusedClosest++;
                int ri = __float_as_int(hit.w);
                toComplex = ri >= 0 && at32(sg.rec_class, (uint32_t)ri) != 0;
                if (toComplex)
                    slotF4(st, A_HIT, slot) = hit;           // shaded by the class-1 launch that follows
            } else {
                hit = slotF4(st, A_HIT, slot);
            }
          if (!toComplex) {
            float4 em4 = slotF4(st, A_EMI, slot);
            em = xyz(em4);
            const int hitInst = ((M & FEAT_INSTANCES) && s.num_instances) ? __float_as_int(em4.w) : -1;   // written by k_trace_closest<.., INST>
            uint4 misc = slotU4(st, A_MISC, slot);
            uint2 rs = make_uint2(misc.x, misc.y);
            uint32_t pixel = misc.z;
            Rng rng;
            rng.state = ((uint64_t)rs.y << 32) | rs.x;
            rng.inc = ((uint64_t)pixel << 1) | 1u;
            rng.sobol = nullptr;
            rng.scramble = rng.index = rng.dim = 0u;
            if ((M & FEAT_QMC) && (pp.flags & TGHIP_PASS_SOBOL)) {
                uint4 sm = slotU4(st, A_SAMP, slot);             // .x = sample index, .z = next dimension
                rngStartSobol(rng, s, pp, pixel % pp.width, pixel/pp.width, pixel, sm.x, sm.z);
            }
            RayD ray;
            ray.o = xyz(ro); ray.d = xyz(rd); ray.tmin = ro.w; ray.tmax = rd.w;
            f3 throughput = xyz(thr4);
            uint32_t flags = __float_as_uint(thr4.w);
            int bounce = (int)FLAG_BOUNCE(flags);
            bool wasSpecular = (flags & FLAG_SPECULAR) != 0;
            uint32_t state = ST_ACTIVE;

            // auxiliary output buffers (TGHIP_PASS_AUX; PathTracer.cpp:46-47, 78-96, 133-140)
            const bool auxOn = (M & FEAT_AUX) && (pp.flags & TGHIP_PASS_AUX) != 0u;
            bool recorded = auxOn && (flags & FLAG_AUX_RECORDED) != 0u;   // recordedOutputValues
            bool auxStore = false;                           // aux0 / aux1 changed
            int loopExit = 0;                                // the while loop was left: 1 = by `break` (bounce not advanced), 2 = bounce limit
            float4 aux0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), aux1 = aux0;
            if constexpr ((M & FEAT_AUX) != 0u) {
                if (auxOn && !recorded) { aux0 = slotF4(st, A_AUX0, slot); aux1 = slotF4(st, A_AUX1, slot); }
            }
            // loop epilogue (PathTracer.cpp:108-126) for a path that goes on from `o` in direction `d`
            auto continuePath = [&](f3 o, f3 d, float tmin) {
                ray.o = o; ray.d = d; ray.tmin = tmin; ray.tmax = PT_INF;
                if (max3(throughput) == 0.0f) {
                    state = ST_TERMINATED;                   // the env term after `break` is throughput*L = 0
                    if constexpr ((M & FEAT_AUX) != 0u) loopExit = 1;
                } else {
                    float roulettePdf = fmaxf(fabsf(throughput.x), fmaxf(fabsf(throughput.y), fabsf(throughput.z)));
                    bool killed = false;
                    if (bounce > 2 && roulettePdf < 0.1f) {
                        if (rngNextBoolean(rng, roulettePdf))
                            throughput = throughput/roulettePdf;
                        else
                            killed = true;
                    }
                    if (killed) {
                        state = ST_TERMINATED;
                    } else if (isnan(sum3(ray.d) + sum3(ray.o)) || isnan(sum3(throughput) + sum3(em))) {
                        state = ST_TERMINATED_BLACK;
                    } else {
                        bounce++;
                        state = bounce < maxBounces ? ST_ACTIVE : ST_TERMINATED;
                        if constexpr ((M & FEAT_AUX) != 0u) { if (state != ST_ACTIVE) loopExit = 2; }
                    }
                }
                if constexpr ((M & FEAT_AUX) != 0u) {
                    if (auxOn && !recorded && loopExit) {    // the sample leaves the loop without having recorded its output values
                        auxPostLoop<M>(s, ray.d, loopExit == 1 && bounce >= minBounces && bounce < maxBounces && s.num_infinite_lights > 0, bounce, aux0, aux1);
                        recorded = true; auxStore = true;
                    }
                }
            };

            // participating media (PathTracer.cpp:48-61): a path inside a medium samples a distance along the segment first
            int med = -1;
            uint32_t medBounce = 0;
            bool volumeEvent = false, mediumEnd = false;
            f3 volP = splat3(0.0f);
            if (M & FEAT_MEDIA) {
                med = FLAG_MEDIUM(flags);
                medBounce = FLAG_MEDIUM_BOUNCE(flags);
                if (med >= 0) {
                    f3 w; float t; bool exited;
                    if (!mediumSampleDistance<M>(s, med, rng, __float_as_int(hit.w) >= 0 ? hit.x : PT_INF, medBounce, w, t, exited)) {
                        mediumEnd = true;                    // "return emission"
                    } else {
                        throughput = throughput*w;           // mediumSample.emission = 0
                        volumeEvent = !exited;
                        volP = ray.o + ray.d*t;
                    }
                }
            }

            if ((M & FEAT_MEDIA) && mediumEnd) {
                state = ST_TERMINATED;
            } else if ((M & FEAT_MEDIA) && volumeEvent) {
                // TraceBase::handleVolume (TraceBase.cpp:496-514) with volumeEstimateDirect / volumeSampleDirect /
                // volumeLightSample / volumePhaseSample (:323-381, 402-414, 471-481)
                const TgHipMedium &mm = s.media[med];
                const bool volumeNee = s.settings.enable_volume_light_sampling != 0;
                wasSpecular = !volumeNee;
                if (volumeNee && bounce < maxBounces - 1) {
                    float lightWeight = 1.0f;
                    int light = chooseLight<M>(s, rng, volP, lightWeight);
                    if (light >= 0) {
                        const uint32_t tag = SHADOW_TAG_MEDIA(light, med, bounce + 1);   // the medium is not re-selected at a volume vertex
                        bool q0 = false, q1 = false;
                        const bool meshLight = s.objects[light].type == TGHIP_OBJ_MESH;
                        const bool diracLight = s.objects[light].type == TGHIP_OBJ_POINT;
                        {
                            f3 d; float dist, pdf;
                            if (lightSampleDirect<M>(s, light, volP, rng, d, dist, pdf)) {
                                float f = phaseEval(mm, ray.d, d);
                                if (f != 0.0f && meshLight) {
                                    float k = powerHeuristic(pdf, f)/pdf;                  // phase pdf == phase value
                                    slotF4(st, A_SH_D0, slot) = mk4(d, dist);
                                    slotF4(st, A_SH_C0, slot) = mk4(splat3(f*k), __uint_as_float(tag));
                                    q0 = true;
                                } else if (f != 0.0f) {
                                    RayD sr; sr.o = volP; sr.d = d; sr.tmin = 0.0f; sr.tmax = PT_INF;   // parentRay.scatter(p, d, 0.0f)
                                    LightHit lh;
                                    bool reached;
                                    if (diracLight) { lh.t = dist; lh.u = 0.0f; lh.v = 0.0f; lh.backSide = false; lh.n = splat3(0.0f); reached = true; }
                                    else reached = lightIntersect<M>(s, light, sr, lh) && !(lh.t*(1.0f + 1e-3f) < dist);
                                    if (reached) {
                                        f3 e = lightEvalDirect<M>(s, light, lh.u, lh.v, lh.backSide);
                                        if (!isZero(e)) {
                                            f3 lightF = e*f/pdf;
                                            if (!diracLight)
                                                lightF = lightF*powerHeuristic(pdf, f);
                                            slotF4(st, A_SH_D0, slot) = mk4(d, lh.t);
                                            slotF4(st, A_SH_C0, slot) = mk4(lightF, __uint_as_float(tag));
                                            q0 = true;
                                        }
                                    }
                                }
                            }
                        }
                        if (!diracLight) {
                            f3 w; float ppdf;
                            phaseSample<M>(mm, rng, ray.d, w, ppdf);
                            if (meshLight) {
                                slotF4(st, A_SH_D1, slot) = mk4(w, ppdf);                      // directPdf needs the hit
                                slotF4(st, A_SH_C1, slot) = mk4(splat3(1.0f), __uint_as_float(tag));
                                q1 = true;
                            } else {
                                RayD sr; sr.o = volP; sr.d = w; sr.tmin = 0.0f; sr.tmax = PT_INF;
                                LightHit lh;
                                if (lightIntersect<M>(s, light, sr, lh)) {
                                    f3 e = lightEvalDirect<M>(s, light, lh.u, lh.v, lh.backSide);
                                    if (!isZero(e)) {
                                        f3 phaseF = e*powerHeuristic(ppdf, lightDirectPdf<M>(s, light, w, volP, lh));
                                        slotF4(st, A_SH_D1, slot) = mk4(w, lh.t);
                                        slotF4(st, A_SH_C1, slot) = mk4(phaseF, __uint_as_float(tag));
                                        q1 = true;
                                    }
                                }
                            }
                        }
                        if (q0 || q1) {
                            hasShadow = true;
                            if (!q0) slotF4(st, A_SH_C0, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                            if (!q1) slotF4(st, A_SH_C1, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                            slotF4(st, A_SH_O, slot) = mk4(volP, 0.0f);
                            slotF4(st, A_SH_W, slot) = mk4(throughput, lightWeight);
                        }
                    }
                }
                f3 w; float ppdf;
                phaseSample<M>(mm, rng, ray.d, w, ppdf);         // the continuation; throughput *= 1
                continuePath(volP, w, 0.0f);
            } else if (__float_as_int(hit.w) < 0) {
                // path escaped: TraceBase::handleInfiniteLights (TraceBase.cpp:570-578); the last infinite light wins
                if ((M & FEAT_INFINITE) && bounce >= minBounces && bounce < maxBounces && s.num_infinite_lights > 0) {
                    // intersectInfinites (TraceableScene.hpp:194-209): every infinite light is asked, the last hit stays
                    int objIdx = -1;
                    for (uint32_t li = 0; li < s.num_infinite_lights; ++li) {
                        const TgHipObject &c = s.objects[s.infinite_lights[li]];
                        if (c.type != TGHIP_OBJ_INFINITE_SPHERE_CAP || dot(ray.d, ld3(c.normal)) >= c.scale[0])
                            objIdx = s.infinite_lights[li];
                    }
                    if (objIdx >= 0) {
                        const TgHipObject &o = s.objects[objIdx];
                        if (!nee || wasSpecular || !(o.flags & TGHIP_OBJF_SAMPLE)) {
                            float u = 0.0f, v = 0.0f, sinTheta;
                            if (o.type == TGHIP_OBJ_INFINITE_SPHERE) infDirectionToUV(o, ray.d, u, v, sinTheta);
                            em = em + throughput*textureEval<M>(s, o.emission, u, v);
                        }
                    }
                }
                state = isnan(sum3(throughput) + sum3(em)) ? ST_TERMINATED_BLACK : ST_TERMINATED;
                if constexpr ((M & FEAT_AUX) != 0u) {
                    if (auxOn && !recorded && state == ST_TERMINATED) {   // (a NaN sample returns before the block at :133)
                        auxPostLoop<M>(s, ray.d, bounce >= minBounces && bounce < maxBounces && s.num_infinite_lights > 0, bounce, aux0, aux1);
                        recorded = true; auxStore = true;
                    }
                }
            } else {
                PROF(1);
                if constexpr ((M & FEAT_AUX) != 0u) {
                    if (auxOn && !recorded) { aux0.w += hit.x; auxStore = true; }   // hitDistance += ray.farT() (PathTracer.cpp:64)
                }
                Info info;
                intersectionInfo<M>(s, ray, hit, info, hitInst);
                const uint32_t lobes = s.bsdfs[info.bsdf].lobes;
                PROF(2);

                // TraceBase::makeLocalScatterEvent (TraceBase.cpp:24-51)
                Frame frame = frameFromNormal(info.Ns);
                bool hitBackside = dot(frame.normal, ray.d) > 0.0f;
                bool flipped = s.settings.enable_two_sided_shading && hitBackside && !(lobes & LOBE_TRANSMISSIVE);
                if (flipped) {
                    frame.normal = -frame.normal;
                    frame.tangent = -frame.tangent;
                }
                Event ev;
                ev.wi = toLocal(frame, -ray.d);
                ev.u = info.u; ev.v = info.v; ev.rng = &rng;
                const bool consistency = s.settings.enable_consistency_checks != 0;
                auto isConsistent = [&](f3 woLocal, f3 w) {      // TraceBase.cpp:53-60
                    if (!consistency) return true;
                    bool geometricBackside = dot(w, info.Ng) < 0.0f;
                    bool shadingBackside = (woLocal.z < 0.0f) != flipped;
                    return geometricBackside == shadingBackside;
                };

                bool auxVisPending = false;
                f3 transparency = splat3(0.0f);
                if (lobes & TGHIP_LOBE_FORWARD) {
                    ev.wo = -ev.wi; ev.requested = TGHIP_LOBE_FORWARD;
                    transparency = bsdfEval<M>(s, info.bsdf, ev);
                }
                float transparencyScalar = avg3(transparency);
                f3 wo;
                bool alive = true;
                if (rngNextBoolean(rng, transparencyScalar)) {
                    wo = ray.d;
                    throughput = throughput*(transparency/transparencyScalar);
                } else {
                    f3 pending = splat3(0.0f);
                    // ---- next-event estimation: TraceBase::estimateDirect (TraceBase.cpp:483-494) ----
                    if (nee && bounce < maxBounces - 1) {
                        float lightWeight = 1.0f;
                        int light = chooseLight<M>(s, rng, info.p, lightWeight);
                        bool pureSpecular = lobes != 0 && (lobes & ~(uint32_t)LOBE_SPECULAR) == 0;
                        if (light >= 0 && !pureSpecular && lobes != TGHIP_LOBE_FORWARD) {
                            uint32_t tag = (uint32_t)light | ((uint32_t)(bounce + 1) << 24);
                            // media scenes: each shadow ray starts in the medium on its side of the surface (TraceBase.cpp:260-261, 302-303)
                            auto mediaTag = [&](f3 dir) {
                                return SHADOW_TAG_MEDIA(light, selectMedium(s.objects[info.object], med, dot(dir, info.Ng) < 0.0f), bounce + 1);
                            };
                            bool q0 = false, q1 = false;
                            f3 inlineResult = splat3(0.0f);
                            const bool meshLight = (M & FEAT_MESHLIGHT) && s.objects[light].type == TGHIP_OBJ_MESH;
                            const bool diracLight = (M & FEAT_SOLIDS) && s.objects[light].type == TGHIP_OBJ_POINT;
                            // lightSample (TraceBase.cpp:246-285)
                            {
                                f3 d; float dist, pdf;
                                if (lightSampleDirect<M>(s, light, info.p, rng, d, dist, pdf)) {
                                    if (M & FEAT_MEDIA) tag = mediaTag(d);
                                    ev.wo = toLocal(frame, d);
                                    ev.requested = LOBE_ALL_BUT_SPECULAR;
                                    if (isConsistent(ev.wo, d)) {
                                        f3 f = bsdfEval<M>(s, info.bsdf, ev);
                                        if ((M & FEAT_MESHLIGHT) && !isZero(f) && meshLight) {
                                            // mesh emitter: whether the ray reaches the light, and with which emission, is only
                                            // known after the scene traversal (TriangleMesh::intersect is a BVH query), so the
                                            // shadow kernel completes f*e/pdf * powerHeuristic from these factors
                                            float k = powerHeuristic(pdf, bsdfPdf<M>(s, info.bsdf, ev))/pdf;
                                            slotF4(st, A_SH_D0, slot) = mk4(d, dist);
                                            slotF4(st, A_SH_C0, slot) = mk4(f*k, __uint_as_float(tag));
                                            q0 = true;
                                        } else if (!isZero(f)) {
                                            RayD sr; sr.o = info.p; sr.d = d; sr.tmin = 5e-4f; sr.tmax = PT_INF;
                                            LightHit lh;
                                            // attenuatedEmission's analytic hit + distance check (TraceBase.cpp:155-162); a Dirac light
                                            // (point) is not intersected: the shadow ray simply ends at the sampled distance
                                            bool reached;
                                            if (diracLight) { lh.t = dist; lh.u = 0.0f; lh.v = 0.0f; lh.backSide = false; lh.n = splat3(0.0f); reached = true; }
                                            else reached = lightIntersect<M>(s, light, sr, lh) && !(lh.t*(1.0f + 1e-3f) < dist);
                                            if (reached) {
                                                f3 e = lightEvalDirect<M>(s, light, lh.u, lh.v, lh.backSide);
                                                // TGHIP_PASS_AUX: attenuatedEmission reports the shadow ray's transmittance even when the
                                                // light shows a black side (TraceBase.cpp:169-170): the ray is traced for the visibility output
                                                bool wantVis = false;
                                                if constexpr ((M & FEAT_AUX) != 0u) wantVis = auxOn && !recorded;
                                                if (!isZero(e) || wantVis) {
                                                    f3 lightF = f*e/pdf;                          // (zero for a black e)
                                                    if (!diracLight)                              // no MIS against a Dirac light (:281-282)
                                                        lightF = lightF*powerHeuristic(pdf, bsdfPdf<M>(s, info.bsdf, ev));
                                                    if (FUSE & FUSE_SHADOW) {
                                                        sr.tmax = lh.t;
                                                        fusedShadow++;
                                                        if (!traverseOccluded<true, true, shadeKinds(M)>(sg, sr, light, nullptr, 0, fusedNodes, fusedPrims) && bounce + 1 >= minBounces)
                                                            inlineResult = inlineResult + lightF;
                                                    } else {
                                                        slotF4(st, A_SH_D0, slot) = mk4(d, lh.t);
                                                        slotF4(st, A_SH_C0, slot) = mk4(lightF, __uint_as_float(tag));
                                                    }
                                                    q0 = true;
                                                }
                                            }
                                        }
                                    }
                                }
                            }
                            // bsdfSample (TraceBase.cpp:287-321); not for Dirac lights (:396-397)
                            if (!diracLight) {
                                ev.requested = LOBE_ALL_BUT_SPECULAR;
                                ev.weight = splat3(1.0f); ev.pdf = 1.0f;
                                if (bsdfSample<M>(s, info.bsdf, ev) && !isZero(ev.weight)) {
                                    f3 wog = toGlobal(frame, ev.wo);
                                    if (M & FEAT_MEDIA) tag = mediaTag(wog);
                                    if ((M & FEAT_MESHLIGHT) && meshLight) {
                                        if (isConsistent(ev.wo, wog)) {
                                            slotF4(st, A_SH_D1, slot) = mk4(wog, ev.pdf);          // directPdf needs the hit
                                            slotF4(st, A_SH_C1, slot) = mk4(ev.weight, __uint_as_float(tag));
                                            q1 = true;
                                        }
                                    } else if (isConsistent(ev.wo, wog)) {
                                        RayD sr; sr.o = info.p; sr.d = wog; sr.tmin = 5e-4f; sr.tmax = PT_INF;
                                        LightHit lh;
                                        if (lightIntersect<M>(s, light, sr, lh)) {
                                            f3 e = lightEvalDirect<M>(s, light, lh.u, lh.v, lh.backSide);
                                            if (!isZero(e)) {
                                                f3 bsdfF = e*ev.weight;
                                                bsdfF = bsdfF*powerHeuristic(ev.pdf, lightDirectPdf<M>(s, light, wog, info.p, lh));
                                                if (FUSE & FUSE_SHADOW) {
                                                    sr.tmax = lh.t;
                                                    fusedShadow++;
                                                    if (!traverseOccluded<true, true, shadeKinds(M)>(sg, sr, light, nullptr, 0, fusedNodes, fusedPrims) && bounce + 1 >= minBounces)
                                                        inlineResult = inlineResult + bsdfF;
                                                } else {
                                                    slotF4(st, A_SH_D1, slot) = mk4(wog, lh.t);
                                                    slotF4(st, A_SH_C1, slot) = mk4(bsdfF, __uint_as_float(tag));
                                                }
                                                q1 = true;
                                            }
                                        }
                                    }
                                }
                            }
                            if ((FUSE & FUSE_SHADOW) && (q0 || q1)) {
                                // emission += estimateDirect(...)*throughput, like k_trace_shadow
                                em = em + (inlineResult*lightWeight)*throughput;
                            } else if (q0 || q1) {
                                hasShadow = true;
                                auxVisPending = q0;
                                if (!q0) slotF4(st, A_SH_C0, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                                if (!q1) slotF4(st, A_SH_C1, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                                slotF4(st, A_SH_O, slot) = mk4(info.p, 5e-4f);
                                slotF4(st, A_SH_W, slot) = mk4(throughput, lightWeight);
                            }
                        }
                    }
                    PROF(3);
                    // emission of the surface itself (TraceBase.cpp:540-543)
                    {
                        const TgHipObject &o = s.objects[info.object];
                        if (o.emission >= 0 && bounce >= minBounces && (!nee || wasSpecular || o.light < 0))
                            pending = lightEvalDirect<M>(s, info.object, info.u, info.v, info.backSide)*throughput;
                    }
                    // with a shadow ray pending, `pending` is added after the NEE term by k_trace_shadow, like the reference
                    if (!hasShadow)
                        em = em + pending;
                    else
                        pendingOut = pending;

                    // continuation: bsdf.sample(event, adjoint = false) with all lobes (TraceBase.cpp:546-558)
                    ev.requested = LOBE_ALL;
                    ev.weight = splat3(1.0f); ev.pdf = 1.0f;
                    if (!bsdfSample<M>(s, info.bsdf, ev)) {
                        alive = false;
                    } else {
                        wo = toGlobal(frame, ev.wo);
                        if (!isConsistent(ev.wo, wo)) {
                            alive = false;
                        } else {
                            throughput = throughput*ev.weight;
                            wasSpecular = (ev.sampled & LOBE_SPECULAR) != 0;
                        }
                    }
                }

                PROF(4);
                if constexpr ((M & FEAT_AUX) != 0u) if (auxOn && !recorded && (!wasSpecular || !alive)) {       // PathTracer.cpp:78-96
                    int ab = info.bsdf;                          // TransparencyBsdf: its base's albedo
                    if (s.bsdfs[ab].type == TGHIP_BSDF_TRANSPARENCY) ab = s.bsdfs[ab].sub0;
                    f3 albedo = textureEval<M>(s, s.bsdfs[ab].albedo, info.u, info.v);
                    if (s.objects[info.object].emission >= 0)    // isEmissive(): + evalDirect
                        albedo = albedo + lightEvalDirect<M>(s, info.object, info.u, info.v, info.backSide);
                    aux0 = mk4(info.Ns, aux0.w);                 // .w = hitDistance
                    // visibility = transmittance of this vertex' light sample, if it got as far as its shadow ray: pending
                    aux1 = mk4(albedo, auxVisPending ? PT_INF : __uint_as_float(0x7FC00000u));
                    recorded = true;
                    auxStore = true;
                }
                if (!alive) {
                    state = ST_TERMINATED;
                } else {
                    f3 hp = ray.o + ray.d*hit.x;                 // ray.hitpoint()
                    if (M & FEAT_MEDIA) {                        // TraceBase.cpp:561-563
                        med = selectMedium(s.objects[info.object], med, dot(wo, info.Ng) < 0.0f);
                        medBounce = 0;                           // state.reset()
                    }
                    continuePath(hp, wo, 5e-4f);
                }
            }
            if (state == ST_ACTIVE) {
                slotF4(st, A_RAY_O, slot) = mk4(ray.o, ray.tmin);
                slotF4(st, A_RAY_D, slot) = mk4(ray.d, ray.tmax);
                *reinterpret_cast<uint2 *>(&slotU4(st, A_MISC, slot)) = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                if ((M & FEAT_QMC) && (pp.flags & TGHIP_PASS_SOBOL))
                    slotU4(st, A_SAMP, slot).z = rng.dim;
            }
            if constexpr ((M & FEAT_AUX) != 0u) if (auxOn) {
                if (!recorded && state != ST_ACTIVE) {       // the sample returned early: it adds nothing (hitDistance is not a depth)
                    aux0.w = __uint_as_float(0x7FC00000u);
                    auxStore = true;
                }
                if (auxStore) { slotF4(st, A_AUX0, slot) = aux0; slotF4(st, A_AUX1, slot) = aux1; }
            }
            const uint32_t newFlags = FLAG_MAKE(bounce, wasSpecular, state) | ((M & FEAT_MEDIA) ? FLAG_MEDIUM_BITS(med, medBounce) : 0u)
                                    | (recorded ? FLAG_AUX_RECORDED : 0u);
            survives = state == ST_ACTIVE;
            black = state == ST_TERMINATED_BLACK;
            if (hasShadow) {
                // k_trace_shadow adds the NEE term, then finishes the path if it ended here
                slotF4(st, A_EMI, slot) = mk4(em, 0.0f);
                slotF4(st, A_SH_P, slot) = mk4(pendingOut, __uint_as_float(newFlags));
            } else if (survives) {
                slotF4(st, A_EMI, slot) = mk4(em, 0.0f);
            } else {
                finished = true;
            }
            if (survives)
                slotF4(st, A_THR, slot) = mk4(throughput, __uint_as_float(newFlags));
          }
        }
        PROF(5);
        if (!DIRECT) {
            if (FUSE & FUSE_TRACE) queuePush(toComplex, local, L, Q_SHADE1);
            queuePush(hasShadow, local, L, Q_SHADOW);
        }
        PROF(6);
        bool regenerated = nextPath<true, (M & FEAT_QMC) != 0>(s, st, pp, finished, false, slot, em, black, &L.cursor, aborted, finishedCount);
        PROF(7);
        if (DIRECT) {
            if (finished && !regenerated) idle |= 1u << turn;   // the work items ran out: nothing left for this slot
        } else {
            queuePush(survives, local, L, Q_EXT);
            queuePush(regenerated, local, L, Q_EXTP);
        }
        PROF(8);
    }
    if (!DIRECT)
        break;
    if (__ballot(idle != 0xFFFFFFFFu) == 0ull)
        break;                                   // every slot of this wave has drained
  }
    if (DIRECT) {
        // nothing is queued any more: the bitmaps go back empty
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < (uint32_t)Q_COUNT*(st.slots_per_block >> 5); w += blockDim.x)
            L.bm[w/(st.slots_per_block >> 5)][w % (st.slots_per_block >> 5)] = 0u;
    }
    PROF_FLUSH(st.stats[blockIdx.x]);
    waveAddStat(&L.samples, finishedCount);
    if (FUSE) {
        waveAddStat(&L.closest_rays, fusedClosest);
        waveAddStat(&L.shadow_rays, fusedShadow);
        waveAddStat(&L.prims, fusedPrims);
    }
    const bool anyExt = queuesEnd(L, st, qIn, appendMask, qIn2);
    if (threadIdx.x == 0) {
        ctl.item_cursor = L.cursor;
        ctl.samples += L.samples;
        if (FUSE) {
            ctl.closest_rays += L.closest_rays; ctl.shadow_rays += L.shadow_rays;
            st.stats[blockIdx.x].prims_tested += L.prims;
            // fused mode has no k_trace_shadow launch: the last shading launch of the iteration reports liveness
            if (anyExt) st.live[0] = (uint32_t)pp.iter_tag;
        }
    }
}

// TraceBase::generalizedShadowRay (TraceBase.cpp:62-125) for the shadow rays queued by k_shade:
// a closest-hit query up to the light; unoccluded iff nothing is hit or the closest hit is the
// light itself (endCap); surfaces with a forward lobe attenuate and the ray continues (FORWARD variant only:
// scenes without a forward-lobe BSDF run the lean variant).
template<bool COUNT, bool FORWARD, bool FLAT, int INST = 0>
__global__ __launch_bounds__(512) void k_trace_shadow(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    extern __shared__ int ldsStack[];
    __shared__ BlockLds L;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    queuesBegin(L, st, ctl, Q_SHADOW, (1u << Q_EXT) | (1u << Q_EXTP), reinterpret_cast<unsigned short *>(ldsStack));
    const uint32_t n = L.n;
    const OrderRegs ord = orderPreload(reinterpret_cast<unsigned short *>(ldsStack), n);
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const bool aborted = st.live[1] != 0;
    uint32_t nodes = 0, prims = 0, rays = 0, slots = 0, finishedCount = 0;
    for (uint32_t base = 0, k = 0; base < n; base += blockDim.x, ++k) {
        uint32_t i = base + threadIdx.x;
        uint32_t slot = 0, local = 0;
        bool finished = false, black = false;
        f3 em = splat3(0.0f);
        if (i < n) {
            local = orderGet(ord, k);
            slot = first + local;
            slots++;
            float4 so = slotF4(st, A_SH_O, slot);
            f3 result = splat3(0.0f);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float4 c = r == 0 ? slotF4(st, A_SH_C0, slot) : slotF4(st, A_SH_C1, slot);
                uint32_t tag = __float_as_uint(c.w);
                if (tag == 0xFFFFFFFFu)
                    continue;
                float4 sd = r == 0 ? slotF4(st, A_SH_D0, slot) : slotF4(st, A_SH_D1, slot);
                int endCap = (int)(tag & 0xFFFFFFu);
                int bounce = (int)(tag >> 24);
                int medium = -1;                               // media scenes (always the FORWARD walk): SHADOW_TAG_MEDIA
                if (FORWARD && s.num_media) { endCap = (int)(tag & 0xFFFFu); medium = (int)((tag >> 16) & 0xFFu) - 1; }
                bool startsOnSurface = so.w != 0.0f;           // shadow rays of a volume vertex start at tmin = 0 (parentRay.scatter(p, d, 0.0f))
                RayD ray;
                ray.o = xyz(so); ray.d = xyz(sd); ray.tmin = so.w; ray.tmax = sd.w;
                float remaining = ray.tmax;
                f3 transmittance = splat3(1.0f);
                bool visValid = true;                          // TGHIP_PASS_AUX: attenuatedEmission got as far as its shadow ray
                f3 shadowT = splat3(0.0f);                     // ... whose result this is (before the emission is applied)
                if (!FORWARD) {
                    // no surface of this scene lets light through: any occluder ends the query
                    rays++;
                    if ((INST ? traverseOccludedInst<COUNT, INST == 2 ? KINDS_MESH : KINDS_ALL>(s, ray, endCap, ldsStack + threadIdx.x, blockDim.x, nodes, prims)
                              : traverseOccluded<COUNT, FLAT>(s, ray, endCap, ldsStack + threadIdx.x, blockDim.x, nodes, prims))
                        || bounce < s.settings.min_bounces)
                        transmittance = splat3(0.0f);
                } else {
                const bool meshLight = s.objects[endCap].type == TGHIP_OBJ_MESH;
                f3 meshFactor = splat3(0.0f);                  // mesh emitters: e (light ray) or e*powerHeuristic (bsdf ray)
                float travelled = 0.0f;
                if (meshLight) { ray.tmax = PT_INF; remaining = PT_INF; }   // sd.w carries the expected distance / the bsdf pdf
                for (;;) {
                    int hitInst = -1;
                    float4 hit = INST ? traverseClosestInst<COUNT, INST == 2 ? KINDS_MESH : KINDS_ALL>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims, hitInst)
                                      : traverseClosest<COUNT, FLAT>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims);
                    rays++;
                    int ri = __float_as_int(hit.w);
                    int hitObject = -1;
                    if (ri >= 0)      // geometry reached through an instance belongs to the `instances` primitive (never a light)
                        hitObject = (int)TGHIP_REC_OBJECT(__float_as_uint(at32(s.recs, (uint32_t)(hitInst >= 0 ? hitInst : ri)*3u).w));
                    if (meshLight && ri < 0) { transmittance = splat3(0.0f); visValid = false; break; }   // the ray never reaches the mesh
                    if (medium >= 0)                             // TraceBase.cpp:103-112: ray.farT() is the hit distance when anything was hit
                        transmittance = transmittance*mediumTransmittance(s, medium, ri >= 0 ? hit.x : ray.tmax, startsOnSurface, true);
                    if (ri < 0 || hitObject == endCap) {
                        if (bounce < s.settings.min_bounces) transmittance = splat3(0.0f);
                        if (meshLight) {
                            // attenuatedEmission on a mesh light (TraceBase.cpp:144-174, TriangleMesh.cpp:344-355,469-473,493-496)
                            Info li;
                            intersectionInfo<BSDF_MASK_ALL>(s, ray, hit, li);
                            const TgHipObject &lo = s.objects[endCap];
                            f3 e = lightEvalDirect<BSDF_MASK_ALL>(s, endCap, li.u, li.v, li.backSide);
                            float total = travelled + hit.x;
                            if (r == 0) {
                                if (total*(1.0f + 1e-3f) < sd.w) { e = splat3(0.0f); visValid = false; }   // a nearer part of the mesh than the sampled point
                                meshFactor = e;
                            } else {
                                float directPdf = lengthSq(xyz(so) - li.p)/(-dot(ray.d, li.Ng)*lo.area);
                                meshFactor = e*powerHeuristic(sd.w, directPdf);
                            }
                        }
                        break;
                    }
                    Info info;
                    intersectionInfo<BSDF_MASK_ALL>(s, ray, hit, info, hitInst);
                    const uint32_t lobes = s.bsdfs[info.bsdf].lobes;
                    if (!(lobes & TGHIP_LOBE_FORWARD)) { transmittance = splat3(0.0f); break; }
                    Frame frame = frameFromNormal(info.Ns);
                    bool hitBackside = dot(frame.normal, ray.d) > 0.0f;
                    if (s.settings.enable_two_sided_shading && hitBackside && !(lobes & LOBE_TRANSMISSIVE)) {
                        frame.normal = -frame.normal;
                        frame.tangent = -frame.tangent;
                    }
                    Event fe;
                    fe.wi = toLocal(frame, -ray.d); fe.wo = -fe.wi;
                    fe.requested = TGHIP_LOBE_FORWARD; fe.u = info.u; fe.v = info.v; fe.rng = nullptr;
                    f3 transparency = bsdfEval<FORWARD ? BSDF_MASK_ALL : 0u>(s, info.bsdf, fe);
                    if (isZero(transparency)) { transmittance = splat3(0.0f); break; }
                    transmittance = transmittance*transparency;
                    bounce++;
                    if (bounce >= s.settings.max_bounces) { transmittance = splat3(0.0f); break; }
                    if (s.num_media)                             // :115-116
                        medium = selectMedium(s.objects[info.object], medium, !info.backSide);
                    startsOnSurface = true;
                    ray.o = ray.o + ray.d*hit.x;
                    travelled += hit.x;
                    remaining -= hit.x;
                    ray.tmin = 5e-4f;
                    ray.tmax = remaining;
                }
                shadowT = transmittance;
                if (meshLight) transmittance = transmittance*meshFactor;
                }
                if (!FORWARD) shadowT = transmittance;
                if (r == 0 && (pp.flags & TGHIP_PASS_AUX)) {   // the visibility output of the vertex that recorded (PathTracer.cpp:93-94)
                    float4 &a1 = slotF4(st, A_AUX1, slot);
                    if (isinf(a1.w))
                        a1.w = visValid ? avg3(shadowT) : __uint_as_float(0x7FC00000u);
                }
                if (!isZero(transmittance))
                    result = result + xyz(c)*transmittance;
            }
            float4 w = slotF4(st, A_SH_W, slot);
            float4 p = slotF4(st, A_SH_P, slot);
            em = xyz(slotF4(st, A_EMI, slot));
            em = em + (result*w.w)*xyz(w);                       // emission += estimateDirect(...)*throughput
            em = em + xyz(p);
            uint32_t state = FLAG_STATE(__float_as_uint(p.w));
            if (state == ST_ACTIVE) {
                slotF4(st, A_EMI, slot) = mk4(em, 0.0f);
            } else {
                finished = true;
                black = state == ST_TERMINATED_BLACK;
            }
        }
        bool regenerated = nextPath(s, st, pp, finished, false, slot, em, black, &L.cursor, aborted, finishedCount);
        queuePush(regenerated, local, L, Q_EXTP);
    }
    waveAddStat(&L.samples, finishedCount);
    waveAddStat(&L.shadow_rays, rays);
    waveAddStat(&L.shadow_slots, slots);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }
    const bool anyExt = queuesEnd(L, st, Q_SHADOW, (1u << Q_EXT) | (1u << Q_EXTP));
    if (threadIdx.x == 0) {
        ctl.item_cursor = L.cursor;
        ctl.samples += L.samples; ctl.shadow_rays += L.shadow_rays; ctl.shadow_slots += L.shadow_slots;
        if (COUNT) {
            BlockStats &bs = st.stats[blockIdx.x];
            bs.nodes_visited += L.nodes; bs.prims_tested += L.prims;
            bs.nodes_visited_shadow += L.nodes; bs.prims_tested_shadow += L.prims;
        }
        // last kernel of the iteration: tell the host whether any extension queue still holds work
        if (anyExt) st.live[0] = iterTag;
    }
}

// Shadow rays of BVH scenes without forward-lobe BSDFs, with dynamic fetch like k_trace_closest_dyn: the unit of
// work is a shadow slot (<= 2 any-hit rays, traced one after the other); idle lanes take the workgroup's next slots
// once fewer than 3/4 of the wave is busy.  A finished slot adds the NEE term to its path's radiance; paths that
// had ended at that vertex go to the Q_FIN queue and are finalised + regenerated by k_finish, a launch of its own:
// nextPath (camera ray, filter table, item bookkeeping) needs 112 VGPRs, the traversal loop 82 -- kept apart, this
// kernel runs 5 waves per SIMD instead of 4.
// Dynamic LDS: [expanded queue, 2 B per slot][node stacks, bvhDepth ints per thread].
#ifndef SHADOW_DYN_BOUNDS
#define SHADOW_DYN_BOUNDS __launch_bounds__(512)
#endif
template<bool COUNT, bool SOLIDS = true>
__global__ SHADOW_DYN_BOUNDS void k_trace_shadow_dyn(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    int *stack = ldsDyn + PT_MAX_SLOTS_PER_BLOCK/2 + threadIdx.x;
    const int stride = (int)blockDim.x;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) fetchNext = 0;
    queuesBegin(L, st, ctl, Q_SHADOW, 1u << Q_FIN, order);
    const uint32_t n = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const int minBounces = s.settings.min_bounces;
    uint32_t nodes = 0, prims = 0, rays = 0, slots = 0;

    bool busy = false;
    uint32_t slot = 0, local = 0;
    int r = 0;                                   // ray of the slot being traced (0: light sample, 1: bsdf sample)
    f3 so = splat3(0.0f);
    float eps = 0.0f;
    f3 result = splat3(0.0f);
    RayD ray; ray.o = splat3(0.0f); ray.d = splat3(1.0f); ray.tmin = 0.0f; ray.tmax = 0.0f;
    f3 invD = splat3(1.0f);
    f3 contrib = splat3(0.0f);
    int endCap = -1;
    int cur = 0, sp = 0;
    bool exhausted = false;

    // sets up ray `r` (or the next valid one) of the current slot; returns false when the slot has no ray left
    auto setupRay = [&]() -> bool {
        for (; r < 2; ++r) {
            float4 c = r == 0 ? slotF4(st, A_SH_C0, slot) : slotF4(st, A_SH_C1, slot);
            uint32_t tag = __float_as_uint(c.w);
            if (tag == 0xFFFFFFFFu)
                continue;
            float4 sd = r == 0 ? slotF4(st, A_SH_D0, slot) : slotF4(st, A_SH_D1, slot);
            endCap = (int)(tag & 0xFFFFFFu);
            int bounce = (int)(tag >> 24);
            rays++;
            if (bounce < minBounces)
                continue;                        // contributes nothing (TraceBase.cpp:114-115 with minBounces)
            contrib = xyz(c);
            ray.o = so; ray.d = xyz(sd); ray.tmin = eps; ray.tmax = sd.w;
            invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
            cur = 0; sp = 0;
            return true;
        }
        return false;
    };
    // NEE term -> path radiance; paths that ended at this vertex go on the finished list
    auto finishSlot = [&]() {
        float4 w = slotF4(st, A_SH_W, slot);
        float4 p = slotF4(st, A_SH_P, slot);
        f3 em = xyz(slotF4(st, A_EMI, slot));
        em = em + (result*w.w)*xyz(w);           // emission += estimateDirect(...)*throughput
        em = em + xyz(p);
        slotF4(st, A_EMI, slot) = mk4(em, 0.0f);
        queuePush(FLAG_STATE(__float_as_uint(p.w)) != ST_ACTIVE, local, L, Q_FIN);
        busy = false;
    };

    for (;;) {
        unsigned long long busyMask = __ballot(busy);
        if (!exhausted && __popcll(busyMask) <= 48) {
            unsigned long long want = ~busyMask;
            uint32_t lane = laneId();
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&fetchNext, (uint32_t)__popcll(want));
            base = __shfl(base, 0);
            if (!busy) {
                uint32_t i = base + __popcll(want & ((1ull << lane) - 1ull));
                if (i < n) {
                    local = order[i];
                    slot = first + local;
                    slots++;
                    float4 o4 = slotF4(st, A_SH_O, slot);
                    so = xyz(o4); eps = o4.w;
                    result = splat3(0.0f);
                    r = 0;
                    busy = true;
                    if (!setupRay())
                        finishSlot();
                }
            }
            if (base + (uint32_t)__popcll(want) >= n)
                exhausted = true;
            busyMask = __ballot(busy);
        }
        if (busyMask == 0ull)
            break;
        if (busy) {
            bool pop = true, occluded = false;
            if (cur >= 0) {
                const float4 *nd = &at32(s.nodes, (uint32_t)cur*4u);
                float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
                if (COUNT) nodes++;
                float e0, e1;
                bool h0 = boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ray, invD, ray.tmax, e0);
                bool h1 = boxTest(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), ray, invD, ray.tmax, e1);
                int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                if (h0 && h1) {
                    if (e1 < e0) { stack[sp*stride] = c0; cur = c1; }
                    else { stack[sp*stride] = c1; cur = c0; }
                    sp++;
                    pop = false;
                } else if (h0) { cur = c0; pop = false; }
                else if (h1) { cur = c1; pop = false; }
            } else {
                uint32_t firstRec = TGHIP_LEAF_FIRST(cur), count = TGHIP_LEAF_COUNT(cur);
                for (uint32_t q = firstRec; q < firstRec + count && !occluded; ++q) {
                    if (COUNT) prims++;
                    float tmax = ray.tmax;
                    float4 hit;
                    uint32_t meta;
                    if (testRecord<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, q, ray, tmax, hit, meta) && (int)TGHIP_REC_OBJECT(meta) != endCap)
                        occluded = true;
                }
            }
            bool rayDone = occluded;
            if (!occluded && pop) {
                if (sp == 0) {
                    result = result + contrib;   // nothing in the way: transmittance 1
                    rayDone = true;
                } else {
                    sp--;
                    cur = stack[sp*stride];
                }
            }
            if (rayDone) {
                r++;
                if (!setupRay())
                    finishSlot();
            }
        }
    }
    waveAddStat(&L.shadow_rays, rays);
    waveAddStat(&L.shadow_slots, slots);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }

    queuesEnd(L, st, Q_SHADOW, 1u << Q_FIN);
    if (threadIdx.x == 0) {
        ctl.shadow_rays += L.shadow_rays; ctl.shadow_slots += L.shadow_slots;
        if (COUNT) {
            BlockStats &bs = st.stats[blockIdx.x];
            bs.nodes_visited += L.nodes; bs.prims_tested += L.prims;
            bs.nodes_visited_shadow += L.nodes; bs.prims_tested_shadow += L.prims;
        }
    }
}

// Second half of the dynamic-fetch shadow step: finalises the paths that had ended at the vertex whose shadow rays
// k_trace_shadow_dyn just resolved (Q_FIN), regenerates their slots and reports whether the workgroup has extension
// rays for the next iteration.
__global__ __launch_bounds__(256) void k_finish(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    __shared__ BlockLds L;
    __shared__ unsigned short order[PT_MAX_SLOTS_PER_BLOCK];
    BlockCtl &ctl = st.ctl[blockIdx.x];
    queuesBegin(L, st, ctl, Q_FIN, (1u << Q_EXT) | (1u << Q_EXTP), order);
    const uint32_t nf = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const bool aborted = st.live[1] != 0;
    uint32_t finishedCount = 0;
    for (uint32_t base = 0; base < nf; base += blockDim.x) {
        uint32_t i = base + threadIdx.x;
        bool fin = i < nf;
        uint32_t loc = 0, sl = 0;
        f3 em = splat3(0.0f);
        bool black = false;
        if (fin) {
            loc = order[i];
            sl = first + loc;
            em = xyz(slotF4(st, A_EMI, sl));
            black = FLAG_STATE(__float_as_uint(slotF4(st, A_SH_P, sl).w)) == ST_TERMINATED_BLACK;
        }
        bool regenerated = nextPath(s, st, pp, fin, false, sl, em, black, &L.cursor, aborted, finishedCount);
        queuePush(regenerated, loc, L, Q_EXTP);
    }
    waveAddStat(&L.samples, finishedCount);
    const bool anyExt = queuesEnd(L, st, Q_FIN, (1u << Q_EXT) | (1u << Q_EXTP));
    if (threadIdx.x == 0) {
        ctl.item_cursor = L.cursor;
        ctl.samples += L.samples;
        if (anyExt) st.live[0] = iterTag;
    }
}

// Sums the per-item partial sums of every pixel slot in fixed chunk order into the framebuffer
// (deterministic; no float atomics anywhere on the accumulation path).
__global__ __launch_bounds__(256) void k_resolve(PathState st, PassParams pp, float *fbSum, uint32_t *fbCount)
{
    uint32_t j = blockIdx.x*blockDim.x + threadIdx.x;
    if (j >= pp.pix_slots)
        return;
    uint32_t x, y;
    if (!slotPixel(pp, j, x, y))
        return;
    uint32_t pixel = x + y*pp.width;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    uint32_t cnt = 0;
    for (uint32_t c = 0; c < pp.chunks; ++c) {
        float4 a = at32(st.partial, c*pp.pix_slots + j);
        sx += a.x; sy += a.y; sz += a.z;
        cnt += __float_as_uint(a.w);
    }
    fbSum[(size_t)pixel*3 + 0] += sx;
    fbSum[(size_t)pixel*3 + 1] += sy;
    fbSum[(size_t)pixel*3 + 2] += sz;
    fbCount[pixel] += cnt;
}

// k_resolve of a record pass (gap-free item enumeration, PassParams): one thread per pixel slot of the sorted record list
// sums the pixel's items of this batch in chunk order.
__global__ __launch_bounds__(256) void k_resolve_records(PathState st, PassParams pp, float *fbSum, uint32_t *fbCount)
{
    uint32_t j = blockIdx.x*blockDim.x + threadIdx.x;
    if (j >= pp.num_sorted*16u)
        return;
    uint32_t rec = pp.rec_sorted[j >> 4];
    uint32_t x = (rec % pp.variance_w)*4u + (j & 3u), y = (rec/pp.variance_w)*4u + ((j >> 2) & 3u);
    if (x >= pp.width || y >= pp.height)
        return;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    uint32_t cnt = 0;
    for (uint32_t c = 0; c < pp.num_chunks; ++c) {
        uint32_t start = pp.rec_chunk_start[c];
        if (j >= pp.rec_chunk_start[c + 1u] - start)
            break;                               // chunks only get shorter: the pixel has no further items
        uint32_t w = start + j;
        if (w < pp.item_base || w - pp.item_base >= pp.total_items)
            continue;                            // item of another batch
        float4 a = at32(st.partial, w - pp.item_base);
        sx += a.x; sy += a.y; sz += a.z;
        cnt += __float_as_uint(a.w);
    }
    uint32_t pixel = x + y*pp.width;
    fbSum[(size_t)pixel*3 + 0] += sx;
    fbSum[(size_t)pixel*3 + 1] += sy;
    fbSum[(size_t)pixel*3 + 2] += sz;
    fbCount[pixel] += cnt;
}

// SampleRecord::addSample (path_tracer/SampleRecord.hpp:46-58) over the luminances the pass wrote, one thread per
// record, in the order the reference's renderTile visits them (PathTraceIntegrator.cpp:136-156: the tile's pixels
// row by row, each pixel's samples in index order), so mean and running variance round exactly like the CPU's.
__global__ __launch_bounds__(64) void k_records(PassParams pp, TgHipSampleRecord *records, uint32_t numRecords)
{
    uint32_t r = blockIdx.x*blockDim.x + threadIdx.x;
    if (r >= numRecords)
        return;
    uint32_t rx = r % pp.variance_w, ry = r/pp.variance_w;
    uint32_t tile = (rx >> 2) + (ry >> 2)*pp.tiles_x;
    if (tile % pp.shard_count != pp.shard_index)
        return;
    const uint32_t cnt = pp.rec_count[r], base = pp.rec_lum[r];
    TgHipSampleRecord rec = records[r];
    for (uint32_t py = 0; py < 4u && ry*4u + py < pp.height; ++py)
        for (uint32_t px = 0; px < 4u && rx*4u + px < pp.width; ++px) {
            const float *lum = pp.lum + (size_t)base + (size_t)((py << 2) | px)*cnt;
            for (uint32_t i = 0; i < cnt; ++i) {
                float x = lum[i];
                rec.sample_count++;
                float delta = x - rec.mean;
                rec.mean += delta/(float)rec.sample_count;
                rec.running_variance += delta*(x - rec.mean);
            }
        }
    records[r] = rec;
}

// =============================================================================================
// Host-side shim
// =============================================================================================
namespace {

std::mutex g_errMutex;
std::string g_createError = "no error";

struct DeviceBuffers {
    std::vector<void *> allocs;
    ~DeviceBuffers() { release(); }
    void release() { for (void *p : allocs) (void)hipFree(p); allocs.clear(); }
};

} // namespace

struct tghip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    std::string error = "no error";

    // scene
    bool haveScene = false;
    DeviceBuffers sceneMem;
    DeviceScene scene;
    int bvhDepth = 0;
    uint32_t width = 0, height = 0;

    // framebuffer
    float *fbSum = nullptr;
    uint32_t *fbCount = nullptr;
    float *extSum = nullptr;
    uint32_t *extCount = nullptr;

    // TGHIP_PASS_SOBOL / TGHIP_PASS_RECORDS state
    DeviceBuffers extMem;                 // tile seeds + per-record pass arrays (sized at upload)
    uint32_t *dTileSeeds = nullptr, *dRecIndex = nullptr, *dRecCount = nullptr, *dRecLum = nullptr;
    TgHipSampleRecord *dRecords = nullptr;   // SampleRecords, ceil(W/4) x ceil(H/4)
    float *lum = nullptr;                 // per-sample luminance of the running pass
    size_t lumCap = 0;
    std::vector<uint32_t> hostRecLum, hostRecIndex, hostRecCount, hostSorted, hostChunkStart, hostHint;
    uint32_t *dSorted = nullptr, *dChunkStart = nullptr, *dHint = nullptr;   // gap-free item enumeration of record passes
    size_t sortedCap = 0, chunkStartCap = 0, hintCap = 0;
    std::vector<uint32_t> hostBucket, hostSortTmp;   // counting sort of the owned records by sample count

    // path pool
    DeviceBuffers poolMem;
    PathState pool;
    uint32_t poolSlots = 0;
    uint32_t poolGrid = 0;                // persistent workgroups the pool is laid out for
    uint32_t *hostLive = nullptr;         // pinned mirror of PathState::live for the loop condition
    std::vector<BlockCtl> hostCtl;        // scratch for tghip_get_counters
    std::vector<BlockStats> hostStats;

    // options
    long long maxSlots = 1ll << 21;       // path pool size
    bool maxSlotsSet = false;             // "max_slots" was given explicitly
    long long maxItems = 1ll << 26;       // work items per batch (partial-sum buffer = 16 B each)
    int chunkSamples = 4;                 // samples per work item
    size_t partialCap = 0;
    float4 *partial = nullptr;

    // shading classes of the uploaded scene (rec_class) and the kernel variants chosen for them
    bool haveComplex = false;             // some primitive record uses a class-1 BSDF
    uint32_t complexMask = 0;             // union of the BSDF types inside class-1 materials
    bool haveForward = false;             // some BSDF has a forward lobe (shadow rays attenuate instead of stop)
    bool haveMeshLight = false;           // a triangle mesh is a sampled light: closest-hit shadow walk, MASK_FULL shading
    bool thinlens = false;                // thin-lens camera: passes run the EXT kernel variants (PT_PASS_THINLENS)
    bool haveSolids = false;              // cube / sphere / disk records: the dynamic-fetch kernels' SOLIDS variants
    TgHipAuxPixel *dAux = nullptr;        // auxiliary output buffers (allocated by the first TGHIP_PASS_AUX pass)
    bool auxPass = false;                 // the pass being rendered keeps them: BSDF_MASK_ALL shading, no fused / dynamic-fetch shadow kernels
    int thrShadeAll = 256;                // workgroup size of k_shade<BSDF_MASK_ALL> (media scenes, TGHIP_PASS_AUX passes)
    bool haveCylinder = false;            // cylinder primitives: BSDF_MASK_ALL shading (the only FEAT_CYLINDER variant), never fused
    bool haveMedia = false;               // participating media: BSDF_MASK_ALL shading (the only FEAT_MEDIA variant), closest-hit shadow walk, never fused
    bool haveInstances = false;           // instance records: two-level traversal kernels (INST), MASK_FULL shading, never the flat list
    bool leanScene = false;               // no bitmap texture, no infinite light, <= 1 sampled light, no triangles: k_shade<MASK_LEAN>
    bool countTraversal = false;
    int checkInterval = 4;                // wavefront iterations between host-side liveness checks
    int blocksPerCuOpt = 0;               // "blocks_per_cu" option; 0 = auto (see chooseThreads)
    int blocksPerCu = 4;                  // persistent workgroups per CU (the same grid for every kernel of a pass)
    // threads per workgroup, per kernel: chosen at upload so that `blocksPerCu` workgroups of EVERY kernel are
    // resident at once (no second scheduling round), i.e. each kernel runs at its own best occupancy on one grid
    int thrClosest = 256, thrShadow = 256, thrShadeSimple = 192, thrShadeComplex = 128;
    int thrOverride[4] = {0, 0, 0, 0};
    bool loopOpt = true;                  // "run_to_completion": fused flat-list scenes without class-1 materials render in ONE launch
    bool fuseFlatOpt = true;              // "fuse_flat": flat-list scenes without forward lobes trace + shadow-test inside k_shade
    int leafBatch = 1;                    // "leaf_batch" (PathState::leaf_batch)
    long long poolPad = 9472;             // bytes between the per-slot arrays of the pool (multiple of 16)
    bool dynamicFetch = true;             // BVH scenes: closest-hit kernel with dynamic ray fetch (k_trace_closest_dyn)
    bool timeKernels = false;             // HIP events around every launch of the wavefront loop (bench.py roofline)
    std::vector<hipEvent_t> evPool;

    // async pass state
    bool passPending = false;
    int passResult = TGHIP_OK;
    TgHipPassDesc pendingPass;

    // counters
    TgHipCounters counters;
    hipEvent_t evA = nullptr, evB = nullptr;
};

#define HIP_TRY(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (ctx)->error = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return TGHIP_E_HIP;                                                             \
        }                                                                                   \
    } while (0)

template<typename T, typename P>
static int uploadArray(tghip_ctx *ctx, DeviceBuffers &mem, const T *src, size_t count, P *dst)   // P = (restrict-qualified) const T *
{
    size_t bytes = std::max<size_t>(count, 1)*sizeof(T);
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    mem.allocs.push_back(p);
    if (count)
        HIP_TRY(ctx, hipMemcpyAsync(p, src, count*sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *dst = static_cast<const T *>(p);
    return TGHIP_OK;
}

template<typename T>
static int allocArray(tghip_ctx *ctx, DeviceBuffers &mem, size_t count, T **dst)
{
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, std::max<size_t>(count, 1)*sizeof(T)));
    mem.allocs.push_back(p);
    *dst = static_cast<T *>(p);
    return TGHIP_OK;
}

static int bsdfDepth(const TgHipSceneDesc *s, int bi, int depth)
{
    if (bi < 0 || depth > 16) return depth;
    const TgHipBsdf &b = s->bsdfs[bi];
    int d = depth + 1;
    if (b.type == TGHIP_BSDF_SMOOTH_COAT || b.type == TGHIP_BSDF_TRANSPARENCY)
        return bsdfDepth(s, b.sub0, d);
    if (b.type == TGHIP_BSDF_MIXED)
        return std::max(bsdfDepth(s, b.sub0, d), bsdfDepth(s, b.sub1, d));
    return d;
}

// set of BSDF types (bit = 1 << type) in the subtree of bsdf `bi`
static uint32_t bsdfTypeMask(const TgHipSceneDesc *s, int bi, int depth)
{
    if (bi < 0 || uint32_t(bi) >= s->num_bsdfs || depth > 16) return 0;
    const TgHipBsdf &b = s->bsdfs[bi];
    uint32_t m = 1u << uint32_t(b.type);
    if (b.type == TGHIP_BSDF_SMOOTH_COAT || b.type == TGHIP_BSDF_TRANSPARENCY)
        m |= bsdfTypeMask(s, b.sub0, depth + 1);
    if (b.type == TGHIP_BSDF_MIXED)
        m |= bsdfTypeMask(s, b.sub0, depth + 1) | bsdfTypeMask(s, b.sub1, depth + 1);
    return m;
}

// Depth of the subtree under `root` (also validates child references).  Instance records met on the way are collected in
// `instanceRecs` when given; without it (a master's subtree) they are an error, as is an instance sharing its leaf.
static int subtreeDepth(const TgHipSceneDesc *s, int32_t root, size_t &visited, std::vector<uint32_t> *instanceRecs)
{
    std::vector<std::pair<int32_t, int>> stack;
    stack.emplace_back(root, 1);
    int depth = 0;
    while (!stack.empty()) {
        auto cur = stack.back();
        stack.pop_back();
        if (cur.first < 0) {
            uint32_t first = TGHIP_LEAF_FIRST(cur.first), count = TGHIP_LEAF_COUNT(cur.first);
            if (first + count > s->num_recs) return -1;
            for (uint32_t i = first; i < first + count; ++i)
                if (TGHIP_REC_KIND(s->recs[i].meta) == TGHIP_REC_INSTANCE) {
                    if (!instanceRecs || count != 1) return -1;
                    instanceRecs->push_back(i);
                }
            continue;
        }
        if (uint32_t(cur.first) >= s->num_nodes || ++visited > s->num_nodes) return -1;
        depth = std::max(depth, cur.second);
        stack.emplace_back(s->nodes[cur.first].child0, cur.second + 1);
        stack.emplace_back(s->nodes[cur.first].child1, cur.second + 1);
    }
    return depth;
}

// Stack depth the traversal needs: the top-level tree, plus -- with instances -- the deepest master subtree above it.
static int bvhDepthOf(const TgHipSceneDesc *s)
{
    size_t visited = 0;
    std::vector<uint32_t> inst;
    int depth = subtreeDepth(s, 0, visited, &inst);
    if (depth < 0 || inst.size() != s->num_instances) return -1;
    std::vector<uint32_t> roots;
    for (uint32_t i : inst) {
        uint32_t root;
        std::memcpy(&root, &s->recs[i].c[0], 4);
        if (root == 0 || root >= s->num_nodes) return -1;
        roots.push_back(root);
    }
    std::sort(roots.begin(), roots.end());
    roots.erase(std::unique(roots.begin(), roots.end()), roots.end());
    int master = 0;
    for (uint32_t root : roots) {
        int d = subtreeDepth(s, int32_t(root), visited, nullptr);
        if (d < 0) return -1;
        master = std::max(master, d);
    }
    return roots.empty() ? depth : depth + master + 1;
}

static bool isFlat(const tghip_ctx *ctx) { return ctx->scene.num_recs <= TGHIP_FLAT_MAX_RECS && !ctx->haveInstances; }

// Dynamic LDS of the traversal kernels: one node stack of bvhDepth ints per thread (a root-to-leaf walk pushes at
// most one far child per internal level), aliased with the expanded queue (2 B per slot) that is consumed before
// traversal starts.  Flat-list scenes need no stack.
static size_t traceLdsBytes(const tghip_ctx *ctx, int threads)
{
    const bool flat = isFlat(ctx);
    size_t stack = flat ? 0 : size_t(std::max(ctx->bvhDepth, 1))*size_t(threads)*sizeof(int);
    return std::max<size_t>(stack, size_t(PT_MAX_SLOTS_PER_BLOCK)*sizeof(unsigned short));
}

// dynamic-fetch traversal kernels keep the expanded queue next to the stacks
static size_t dynLdsBytes(const tghip_ctx *ctx, int threads)
{
    return size_t(PT_MAX_SLOTS_PER_BLOCK)*sizeof(unsigned short) + size_t(std::max(ctx->bvhDepth, 1))*size_t(threads)*sizeof(int);
}

static int launchGrid(const tghip_ctx *ctx) { return ctx->prop.multiProcessorCount*std::max(ctx->blocksPerCu, 1); }

// Largest workgroup size (multiple of 64, <= maxThreads) at which `blocksPerCu` workgroups of `kernel` fit on a CU.
template<typename K>
static int pickThreads(const tghip_ctx *ctx, K kernel, int maxThreads, int ldsMode)   // 0: no dynamic LDS, 1: traceLdsBytes, 2: dynLdsBytes
{
    for (int t = maxThreads; t >= 128; t -= 64) {
        int nb = 0;
        size_t lds = ldsMode == 1 ? traceLdsBytes(ctx, t) : ldsMode == 2 ? dynLdsBytes(ctx, t) : 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kernel), t, lds) == hipSuccess && nb >= ctx->blocksPerCu)
            return t;
    }
    return 128;
}

// Adds the per-workgroup statistics on the device to ctx->counters and zeroes them there.
static int foldCounters(tghip_ctx *ctx)
{
    if (!ctx->poolSlots)
        return TGHIP_OK;
    const size_t g = ctx->poolGrid;
    ctx->hostCtl.resize(g);
    ctx->hostStats.resize(g);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->hostCtl.data(), ctx->pool.ctl, g*sizeof(BlockCtl), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(ctx->hostStats.data(), ctx->pool.stats, g*sizeof(BlockStats), hipMemcpyDeviceToHost));
    for (size_t b = 0; b < g; ++b) {
        BlockCtl &c = ctx->hostCtl[b];
        ctx->counters.samples += c.samples; ctx->counters.closest_rays += c.closest_rays;
        ctx->counters.shadow_rays += c.shadow_rays; ctx->counters.shadow_slots += c.shadow_slots;
        c.samples = c.closest_rays = c.shadow_rays = c.shadow_slots = 0;
        const BlockStats &t = ctx->hostStats[b];
        ctx->counters.nodes_visited += t.nodes_visited; ctx->counters.prims_tested += t.prims_tested;
        ctx->counters.nodes_visited_shadow += t.nodes_visited_shadow; ctx->counters.prims_tested_shadow += t.prims_tested_shadow;
    }
#ifdef PT_PROFILE
    {
        unsigned long long tot[12] = {0};
        for (size_t b = 0; b < g; ++b) for (int k = 0; k < 12; ++k) tot[k] += ctx->hostStats[b].prof[k];
        unsigned long long sum = 0; for (int k = 0; k < 12; ++k) sum += tot[k];
        if (sum) { std::fprintf(stderr, "[PT_PROFILE] k_shade wave-cycles by section:"); for (int k = 0; k < 9; ++k) std::fprintf(stderr, " s%d=%.1f%%", k, 100.0*double(tot[k])/double(sum)); std::fprintf(stderr, " total=%llu\n", sum); }
    }
#endif
    // no pass is running here (calls on one handle are serialised), so the records can be written back whole
    HIP_TRY(ctx, hipMemcpy(ctx->pool.ctl, ctx->hostCtl.data(), g*sizeof(BlockCtl), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ctx->pool.stats, 0, g*sizeof(BlockStats)));
    return TGHIP_OK;
}

// Pool layout: `grid` persistent workgroups x slotsPerBlock slots (a multiple of 64); queue segments and the
// per-workgroup control records are laid out the same way.
static int ensurePool(tghip_ctx *ctx, uint32_t wantSlots)
{
    const uint32_t grid = uint32_t(launchGrid(ctx));
    uint32_t perBlock = (wantSlots + grid - 1)/grid;
    perBlock = std::min<uint32_t>(PT_MAX_SLOTS_PER_BLOCK, std::max<uint32_t>(64u, (perBlock + 63u)/64u*64u));
    const uint32_t slots = perBlock*grid;
    PathState &p = ctx->pool;
    if (ctx->poolSlots >= slots && ctx->poolGrid == grid) {
        p.num_slots = slots;
        p.slots_per_block = perBlock;
        return TGHIP_OK;
    }
    int rc = foldCounters(ctx);                  // the per-workgroup statistics live in the pool
    if (rc != TGHIP_OK) return rc;
    ctx->poolMem.release();
    ctx->poolSlots = 0;
#define POOL_ALLOC(field, n) do { std::remove_reference<decltype(*p.field)>::type *tmp_ = nullptr; \
        if ((rc = allocArray(ctx, ctx->poolMem, (n), &tmp_)) != TGHIP_OK) return rc; p.field = tmp_; } while (0)
    // arrays are skewed by an odd number of 256-byte units so that element i of different arrays does not map to
    // the same HBM channel (a power-of-two array stride made the kernels' speed depend on allocation luck)
    const uint64_t strideBytes = uint64_t(slots)*16u + uint64_t(ctx->poolPad);
    if (strideBytes*A_COUNT >= (1ull << 32)) { ctx->error = "path pool too large for 32-bit slot offsets"; return TGHIP_E_INVALID; }
    POOL_ALLOC(pool, size_t(strideBytes)*A_COUNT);
    p.stride = uint32_t(strideBytes);
    POOL_ALLOC(bm, size_t(slots/32)*Q_COUNT);
    p.bmStride = slots/32;
    POOL_ALLOC(ctl, grid); POOL_ALLOC(stats, grid); POOL_ALLOC(live, 2);
#undef POOL_ALLOC
    HIP_TRY(ctx, hipMemsetAsync(p.ctl, 0, sizeof(BlockCtl)*grid, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(p.stats, 0, sizeof(BlockStats)*grid, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(p.live, 0, 2*sizeof(uint32_t), ctx->stream));
    p.num_slots = slots;
    p.slots_per_block = perBlock;
    ctx->poolSlots = slots;
    ctx->poolGrid = grid;
    return TGHIP_OK;
}

// Picks the workgroup size of each kernel of the wavefront loop for the uploaded scene (see tghip_ctx::thr*).
static void chooseThreads(tghip_ctx *ctx)
{
    const bool flat = isFlat(ctx);
    // Measured (profiles/README.md): BVH scenes are latency-bound and run best with every workgroup of every
    // kernel resident at once (4 per CU, workgroup size per kernel = that kernel's occupancy limit / 4); flat-list
    // scenes are streaming-bound and prefer 8 small workgroups per CU that the dispatcher load-balances.
    ctx->blocksPerCu = ctx->blocksPerCuOpt > 0 ? ctx->blocksPerCuOpt : (flat ? 8 : 4);
    if (flat && ctx->blocksPerCuOpt == 0) {
        ctx->thrClosest = ctx->thrShadow = ctx->thrShadeSimple = ctx->thrShadeComplex = 256;
    } else {
    const bool inst = ctx->haveInstances;
    const bool dyn = ctx->dynamicFetch && !inst;               // the dynamic-fetch kernels are single-level
    ctx->thrClosest = flat ? pickThreads(ctx, k_trace_closest<false, true>, 512, 1)
                    : inst ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest<false, false, 1>, 512, 1) : pickThreads(ctx, k_trace_closest<false, false, 2>, 512, 1))
                    : dyn ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest_dyn<false, true>, 320, 2) : pickThreads(ctx, k_trace_closest_dyn<false, false>, 320, 2))   // 20 waves/CU measured best (profiles/README.md)
                                        : pickThreads(ctx, k_trace_closest<false, false>, 512, 1);
    if (!flat && !ctx->haveForward && !ctx->haveMeshLight && dyn)
        ctx->thrShadow = ctx->haveSolids ? pickThreads(ctx, k_trace_shadow_dyn<false, true>, 256, 2) : pickThreads(ctx, k_trace_shadow_dyn<false, false>, 256, 2);   // measured: 192 / 256 / 320 / 384 threads = 525 / 462 / 633 / 619 us per launch
    else if (inst)
        ctx->thrShadow = (ctx->haveForward || ctx->haveMeshLight)
                       ? (ctx->haveSolids ? pickThreads(ctx, k_trace_shadow<false, true, false, 1>, 512, 1) : pickThreads(ctx, k_trace_shadow<false, true, false, 2>, 512, 1))
                       : (ctx->haveSolids ? pickThreads(ctx, k_trace_shadow<false, false, false, 1>, 512, 1) : pickThreads(ctx, k_trace_shadow<false, false, false, 2>, 512, 1));
    else if (ctx->haveForward || ctx->haveMeshLight)
        ctx->thrShadow = flat ? pickThreads(ctx, k_trace_shadow<false, true, true>, 512, 1) : pickThreads(ctx, k_trace_shadow<false, true, false>, 512, 1);
    else
        ctx->thrShadow = flat ? pickThreads(ctx, k_trace_shadow<false, false, true>, 512, 1) : pickThreads(ctx, k_trace_shadow<false, false, false>, 512, 1);
    if (ctx->haveMeshLight || inst) ctx->thrShadeSimple = pickThreads(ctx, k_shade<MASK_FULL, 2, 0>, 256, 0);
    else ctx->thrShadeSimple = ctx->leanScene ? pickThreads(ctx, k_shade<MASK_LEAN, LEAN_WAVES, 0>, 256, 0) : pickThreads(ctx, k_shade<MASK_SIMPLE, SIMPLE_WAVES, 0>, 256, 0);
    if (ctx->haveMeshLight || inst)                 ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_FULL, 2, 0>, 256, 0);
    else if ((ctx->complexMask & ~MASK_COAT) == 0)  ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_COAT, COAT_WAVES, 0>, 256, 0);
    else if ((ctx->complexMask & ~MASK_GLASS) == 0) ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_GLASS, 2, 0>, 256, 0);
    else                                            ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_FULL, 2, 0>, 256, 0);
    if (ctx->haveMedia) ctx->thrShadeSimple = ctx->thrShadeComplex = pickThreads(ctx, k_shade<BSDF_MASK_ALL, 2, 0>, 256, 0);
    }
    ctx->thrShadeAll = flat && ctx->blocksPerCuOpt == 0 ? 256 : pickThreads(ctx, k_shade<BSDF_MASK_ALL, 2, 0>, 256, 0);
    int *dst[4] = {&ctx->thrClosest, &ctx->thrShadow, &ctx->thrShadeSimple, &ctx->thrShadeComplex};
    for (int i = 0; i < 4; ++i)
        if (ctx->thrOverride[i] >= 64) *dst[i] = std::min(ctx->thrOverride[i]/64*64, i < 2 ? 512 : 256);
    if (std::getenv("TGHIP_VERBOSE"))
        std::fprintf(stderr, "[tghip] grid %d x threads closest %d shadow %d shade %d/%d (flat %d, forward %d, complex mask 0x%x)\n",
                     launchGrid(ctx), ctx->thrClosest, ctx->thrShadow, ctx->thrShadeSimple, ctx->thrShadeComplex, int(flat), int(ctx->haveForward), ctx->complexMask);
}

extern "C" {

int tghip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

tghip_ctx *tghip_create(int device_ordinal)
{
    int n = tghip_device_count();
    if (device_ordinal < 0 || device_ordinal >= n) {
        std::lock_guard<std::mutex> lock(g_errMutex);
        g_createError = n == 0 ? "no HIP device available" : "device ordinal out of range";
        return nullptr;
    }
    tghip_ctx *ctx = new tghip_ctx();
    ctx->device = device_ordinal;
    std::memset(&ctx->counters, 0, sizeof(ctx->counters));
    std::memset(&ctx->pool, 0, sizeof(ctx->pool));
    std::memset(&ctx->scene, 0, sizeof(ctx->scene));
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device_ordinal);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evA);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evB);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ctx->hostLive), 2*sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> lock(g_errMutex);
        g_createError = std::string("tghip_create: ") + hipGetErrorString(e);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void tghip_destroy(tghip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->sceneMem.release();
    ctx->poolMem.release();
    ctx->extMem.release();
    if (ctx->lum) (void)hipFree(ctx->lum);
    if (ctx->dSorted) (void)hipFree(ctx->dSorted);
    if (ctx->dChunkStart) (void)hipFree(ctx->dChunkStart);
    if (ctx->dHint) (void)hipFree(ctx->dHint);
    if (ctx->fbSum) (void)hipFree(ctx->fbSum);
    if (ctx->fbCount) (void)hipFree(ctx->fbCount);
    if (ctx->dAux) (void)hipFree(ctx->dAux);
    if (ctx->partial) (void)hipFree(ctx->partial);
    if (ctx->hostLive) (void)hipHostFree(ctx->hostLive);
    if (ctx->evA) (void)hipEventDestroy(ctx->evA);
    if (ctx->evB) (void)hipEventDestroy(ctx->evB);
    for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *tghip_last_error(tghip_ctx *ctx)
{
    if (!ctx) {
        std::lock_guard<std::mutex> lock(g_errMutex);
        static thread_local std::string copy;
        copy = g_createError;
        return copy.c_str();
    }
    return ctx->error.c_str();
}

int tghip_set_option(tghip_ctx *ctx, const char *key, long long value)
{
    if (!ctx || !key) return TGHIP_E_INVALID;
    std::string k(key);
    if (k == "count_traversal") ctx->countTraversal = value != 0;
    else if (k == "max_slots") { ctx->maxSlots = std::max<long long>(value, 256); ctx->maxSlotsSet = true; }
    else if (k == "max_items") ctx->maxItems = std::max<long long>(value, 256);
    else if (k == "chunk_samples") ctx->chunkSamples = int(std::min<long long>(std::max<long long>(value, 1), 1 << 20));
    else if (k == "check_interval") ctx->checkInterval = int(std::max<long long>(value, 1));
    else if (k == "time_kernels") ctx->timeKernels = value != 0;
    else if (k == "blocks_per_cu") { ctx->blocksPerCuOpt = int(std::min<long long>(std::max<long long>(value, 0), 8)); if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "leaf_batch") ctx->leafBatch = int(std::min<long long>(std::max<long long>(value, 1), 64));
    else if (k == "fuse_flat") ctx->fuseFlatOpt = value != 0;
    else if (k == "run_to_completion") ctx->loopOpt = value != 0;
    else if (k == "pool_pad") { ctx->poolPad = std::max<long long>(value, 0)/16*16; ctx->poolMem.release(); ctx->poolSlots = 0; }
    else if (k == "dynamic_fetch") { ctx->dynamicFetch = value != 0; if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "threads_closest") { ctx->thrOverride[0] = int(value); if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "threads_shadow") { ctx->thrOverride[1] = int(value); if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "threads_shade_simple") { ctx->thrOverride[2] = int(value); if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "threads_shade_complex") { ctx->thrOverride[3] = int(value); if (ctx->haveScene) chooseThreads(ctx); }
    else { ctx->error = "unknown option '" + k + "'"; return TGHIP_E_INVALID; }
    return TGHIP_OK;
}

int tghip_upload_scene(tghip_ctx *ctx, const TgHipSceneDesc *sd)
{
    if (!ctx || !sd) return TGHIP_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (sd->abi_version != TGHIP_ABI_VERSION) { ctx->error = "scene description ABI version mismatch"; return TGHIP_E_INVALID; }
    if (sd->num_nodes == 0 || !sd->nodes) { ctx->error = "scene has no BVH"; return TGHIP_E_INVALID; }
    if (sd->camera.res_x <= 0 || sd->camera.res_y <= 0) { ctx->error = "invalid camera resolution"; return TGHIP_E_INVALID; }
    if (sd->num_lights > 16) { ctx->error = "more than 16 sampled lights are not supported"; return TGHIP_E_UNSUPPORTED; }
    if (sd->num_objects >= (1u << 24)) { ctx->error = "too many objects"; return TGHIP_E_UNSUPPORTED; }
    if (sd->num_recs >= (1u << 26) || sd->num_nodes >= (1u << 26)) {   // 64-B attribute / node records behind 32-bit byte offsets (at32)
        ctx->error = "more than 2^26 primitive records or BVH nodes are not supported";
        return TGHIP_E_UNSUPPORTED;
    }
    int depth = bvhDepthOf(sd);
    if (depth < 0 || depth > TGHIP_MAX_BVH_DEPTH) { ctx->error = "malformed or too deep BVH"; return TGHIP_E_INVALID; }
    for (uint32_t i = 0; i < sd->num_bsdfs; ++i)
        if (bsdfDepth(sd, int(i), 0) > PT_MAX_BSDF_DEPTH) { ctx->error = "BSDF nesting deeper than 3 is not supported"; return TGHIP_E_UNSUPPORTED; }
    for (uint32_t i = 0; i < sd->num_lights; ++i) {
        int t = sd->objects[sd->lights[i]].type;
        if (t == TGHIP_OBJ_MESH) {
            const TgHipObject &lo = sd->objects[sd->lights[i]];
            if (lo.first_light_tri < 0 || lo.num_light_tris <= 0 || !sd->light_tris ||
                uint64_t(lo.first_light_tri) + uint64_t(lo.num_light_tris)*10u + 1u > sd->num_light_tri_floats) {
                ctx->error = "sampled mesh emitter without a valid light_tris block";
                return TGHIP_E_INVALID;
            }
        } else if (t != TGHIP_OBJ_QUAD && t != TGHIP_OBJ_INFINITE_SPHERE && t != TGHIP_OBJ_CUBE && t != TGHIP_OBJ_SPHERE && t != TGHIP_OBJ_DISK && t != TGHIP_OBJ_INFINITE_SPHERE_CAP && t != TGHIP_OBJ_POINT && t != TGHIP_OBJ_CYLINDER) {
            ctx->error = "unknown emitter type";
            return TGHIP_E_UNSUPPORTED;
        }
    }

    if (ctx->stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sceneMem.release();
    ctx->haveScene = false;
    DeviceScene &s = ctx->scene;
    std::memset(&s, 0, sizeof(s));
    int rc;
    const TgHipBvhNode *dn; const TgHipPrimRec *dr; const TgHipTriAttr *da;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->nodes, sd->num_nodes, &dn)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->recs, sd->num_recs, &dr)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->tri_attrs, sd->num_recs, &da)) != TGHIP_OK) return rc;
    s.nodes = reinterpret_cast<const float4 *>(dn);
    s.recs = reinterpret_cast<const float4 *>(dr);
    s.tri_attrs = reinterpret_cast<const float4 *>(da);
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->objects, sd->num_objects, &s.objects)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->lights, sd->num_lights, &s.lights)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->infinite_lights, sd->num_infinite_lights, &s.infinite_lights)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->bsdfs, sd->num_bsdfs, &s.bsdfs)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->textures, sd->num_textures, &s.textures)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->texels, sd->num_texel_floats, &s.texels)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->dist, sd->num_dist_floats, &s.dist)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->light_tris, sd->num_light_tri_floats, &s.light_tris)) != TGHIP_OK) return rc;
    ctx->haveMeshLight = false;
    ctx->haveInstances = sd->num_instances > 0;
    ctx->haveMedia = sd->num_media > 0;
    ctx->haveCylinder = false;
    for (uint32_t i = 0; i < sd->num_objects; ++i)
        if (sd->objects[i].type == TGHIP_OBJ_CYLINDER) ctx->haveCylinder = true;
    if (ctx->haveMedia) {
        if (!sd->media || sd->num_media > PT_MAX_MEDIA) { ctx->error = "more than 126 media are not supported"; return TGHIP_E_UNSUPPORTED; }
        if (sd->num_objects >= (1u << 16)) { ctx->error = "media scenes support at most 65535 primitives"; return TGHIP_E_UNSUPPORTED; }
        if (sd->camera.medium >= int32_t(sd->num_media)) { ctx->error = "camera medium out of range"; return TGHIP_E_INVALID; }
        for (uint32_t i = 0; i < sd->num_objects; ++i)
            if (sd->objects[i].int_medium >= int32_t(sd->num_media) || sd->objects[i].ext_medium >= int32_t(sd->num_media)) {
                ctx->error = "primitive medium out of range";
                return TGHIP_E_INVALID;
            }
        for (uint32_t i = 0; i < sd->num_media; ++i)
            if (sd->media[i].phase_type < TGHIP_PHASE_ISOTROPIC || sd->media[i].phase_type > TGHIP_PHASE_RAYLEIGH) {
                ctx->error = "unknown phase function";
                return TGHIP_E_UNSUPPORTED;
            }
    }
    ctx->thinlens = sd->camera.type == TGHIP_CAMERA_THINLENS;
    if (sd->camera.type != TGHIP_CAMERA_PINHOLE && sd->camera.type != TGHIP_CAMERA_THINLENS) { ctx->error = "unknown camera type"; return TGHIP_E_UNSUPPORTED; }
    for (uint32_t i = 0; i < sd->num_lights; ++i)
        if (sd->objects[sd->lights[i]].type == TGHIP_OBJ_MESH) ctx->haveMeshLight = true;
    // CDF guide tables for the samplable bitmaps (pt_scene.h: upperBoundGuided)
    {
        std::vector<uint16_t> guide;
        std::vector<int32_t> texGuide(std::max<uint32_t>(sd->num_textures, 1u), -1);
        auto build = [&](const float *a, int n, int buckets) {   // g[b] = upper_bound(a[0..n], b/buckets)
            int idx = 0;
            for (int b = 0; b <= buckets; ++b) {
                float x = float(b)/float(buckets);
                while (idx <= n && a[idx] <= x) ++idx;
                guide.push_back(uint16_t(std::min(idx, n + 1)));
            }
        };
        for (uint32_t i = 0; i < sd->num_textures; ++i) {
            const TgHipTexture &t = sd->textures[i];
            if (t.type != TGHIP_TEX_BITMAP || t.dist_offset < 0 || t.w <= 0 || t.h <= 0 || t.w >= 65535 || t.h >= 65535)
                continue;
            if (guide.size() + size_t(PT_GUIDE_MARGINAL + 1) + size_t(t.h)*(PT_GUIDE_ROW + 1) >= (1u << 31))
                continue;
            texGuide[i] = int32_t(guide.size());
            const float *mpdf = sd->dist + t.dist_offset;
            const float *mcdf = mpdf + t.h;
            const float *cdf = mcdf + (t.h + 1) + size_t(t.w)*t.h;
            build(mcdf, t.h, PT_GUIDE_MARGINAL);
            for (int y = 0; y < t.h; ++y)
                build(cdf + size_t(y)*(t.w + 1), t.w, PT_GUIDE_ROW);
        }
        if (guide.empty()) guide.push_back(0);
        if ((rc = uploadArray(ctx, ctx->sceneMem, guide.data(), guide.size(), &s.guide)) != TGHIP_OK) return rc;
        if ((rc = uploadArray(ctx, ctx->sceneMem, texGuide.data(), texGuide.size(), &s.tex_guide)) != TGHIP_OK) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the vectors go out of scope
    }
    // shading classes ("sort by material"): class 0 = BSDFs made of lambert/null only, class 1 = the rest
    {
        std::vector<uint32_t> typeMask(sd->num_bsdfs, 0u);
        std::vector<uint8_t> recClass(std::max<uint32_t>(sd->num_recs, 1u), 0);
        ctx->haveComplex = false; ctx->complexMask = 0; ctx->haveForward = false; ctx->haveSolids = false;
        for (uint32_t i = 0; i < sd->num_bsdfs; ++i) {
            typeMask[i] = bsdfTypeMask(sd, int(i), 0);
            if (sd->bsdfs[i].lobes & TGHIP_LOBE_FORWARD) ctx->haveForward = true;
        }
        for (uint32_t i = 0; i < sd->num_recs; ++i) {
            uint32_t meta = sd->recs[i].meta;
            if (TGHIP_REC_KIND(meta) == TGHIP_REC_INSTANCE)
                continue;                    // never a hit record itself: hits are the master's triangles
            if (TGHIP_REC_KIND(meta) > TGHIP_REC_CYLINDER) { ctx->error = "unknown primitive record kind"; return TGHIP_E_INVALID; }
            if (TGHIP_REC_KIND(meta) != TGHIP_REC_TRIANGLE && TGHIP_REC_KIND(meta) != TGHIP_REC_QUAD) ctx->haveSolids = true;
            int bi = TGHIP_REC_KIND(meta) == TGHIP_REC_TRIANGLE ? sd->tri_attrs[i].bsdf : sd->objects[TGHIP_REC_OBJECT(meta)].bsdf;
            if (bi < 0 || uint32_t(bi) >= sd->num_bsdfs) { ctx->error = "primitive record without a valid bsdf"; return TGHIP_E_INVALID; }
            bool simple = (typeMask[size_t(bi)] & ~MASK_SIMPLE) == 0 && !(sd->bsdfs[bi].lobes & TGHIP_LOBE_FORWARD);
            recClass[i] = simple ? 0 : 1;
            if (!simple) { ctx->haveComplex = true; ctx->complexMask |= typeMask[size_t(bi)]; }
        }
        bool lean = sd->num_infinite_lights == 0 && sd->num_lights <= 1;
        for (uint32_t i = 0; i < sd->num_textures && lean; ++i) lean = sd->textures[i].type != TGHIP_TEX_BITMAP;
        for (uint32_t i = 0; i < sd->num_recs && lean; ++i)
            lean = TGHIP_REC_KIND(sd->recs[i].meta) == TGHIP_REC_QUAD || TGHIP_REC_KIND(sd->recs[i].meta) == TGHIP_REC_CUBE;
        for (uint32_t i = 0; i < sd->num_lights && lean; ++i) lean = sd->objects[sd->lights[i]].type == TGHIP_OBJ_QUAD;
        ctx->leanScene = lean;
        if ((rc = uploadArray(ctx, ctx->sceneMem, recClass.data(), recClass.size(), &s.rec_class)) != TGHIP_OK) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // recClass goes out of scope
    }
    s.num_nodes = sd->num_nodes; s.num_recs = sd->num_recs; s.num_objects = sd->num_objects;
    s.num_lights = sd->num_lights; s.num_infinite_lights = sd->num_infinite_lights;
    s.num_bsdfs = sd->num_bsdfs; s.num_textures = sd->num_textures;
    s.num_instances = sd->num_instances;
    s.media = nullptr;
    s.num_media = sd->num_media;
    if (sd->num_media && (rc = uploadArray(ctx, ctx->sceneMem, sd->media, size_t(sd->num_media), &s.media)) != TGHIP_OK) return rc;
    if (ctx->haveMedia) ctx->haveForward = true;   // shadow rays pick up transmittance segment by segment: the closest-hit walk
    if ((rc = uploadArray(ctx, ctx->sceneMem, &sd->camera, 1, &s.camera)) != TGHIP_OK) return rc;
    s.sobol = nullptr;
    if (sd->sobol_matrices) {
        if (sd->num_sobol_words != uint64_t(TGHIP_SOBOL_DIMS)*TGHIP_SOBOL_BITS) { ctx->error = "sobol_matrices must hold 1024 x 52 words"; return TGHIP_E_INVALID; }
        if ((rc = uploadArray(ctx, ctx->sceneMem, sd->sobol_matrices, size_t(sd->num_sobol_words), &s.sobol)) != TGHIP_OK) return rc;
    }
    s.settings = sd->settings;
    ctx->bvhDepth = depth;

    // framebuffer
    if (ctx->width != uint32_t(sd->camera.res_x) || ctx->height != uint32_t(sd->camera.res_y) || !ctx->fbSum) {
        if (ctx->fbSum) (void)hipFree(ctx->fbSum);
        if (ctx->fbCount) (void)hipFree(ctx->fbCount);
        if (ctx->dAux) (void)hipFree(ctx->dAux);
        ctx->fbSum = nullptr; ctx->fbCount = nullptr; ctx->dAux = nullptr;
        ctx->width = uint32_t(sd->camera.res_x); ctx->height = uint32_t(sd->camera.res_y);
        size_t npix = size_t(ctx->width)*ctx->height;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->fbSum), npix*3*sizeof(float)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->fbCount), npix*sizeof(uint32_t)));
        // tile seeds, SampleRecords and the per-record pass arrays
        ctx->extMem.release();
        const size_t tiles = size_t((ctx->width + 15)/16)*((ctx->height + 15)/16);
        const size_t recs = size_t((ctx->width + 3)/4)*((ctx->height + 3)/4);
        if ((rc = allocArray(ctx, ctx->extMem, tiles, &ctx->dTileSeeds)) != TGHIP_OK) return rc;
        if ((rc = allocArray(ctx, ctx->extMem, recs, &ctx->dRecIndex)) != TGHIP_OK) return rc;
        if ((rc = allocArray(ctx, ctx->extMem, recs, &ctx->dRecCount)) != TGHIP_OK) return rc;
        if ((rc = allocArray(ctx, ctx->extMem, recs, &ctx->dRecLum)) != TGHIP_OK) return rc;
        if ((rc = allocArray(ctx, ctx->extMem, recs, &ctx->dRecords)) != TGHIP_OK) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->haveScene = true;
    chooseThreads(ctx);
    return tghip_clear_framebuffer(ctx);
}

int tghip_clear_framebuffer(tghip_ctx *ctx)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t npix = size_t(ctx->width)*ctx->height;
    float *sum = ctx->extSum ? ctx->extSum : ctx->fbSum;
    uint32_t *cnt = ctx->extCount ? ctx->extCount : ctx->fbCount;
    HIP_TRY(ctx, hipMemsetAsync(sum, 0, npix*3*sizeof(float), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(cnt, 0, npix*sizeof(uint32_t), ctx->stream));
    const size_t recs = size_t((ctx->width + 3)/4)*((ctx->height + 3)/4);
    HIP_TRY(ctx, hipMemsetAsync(ctx->dRecords, 0, recs*sizeof(TgHipSampleRecord), ctx->stream));
    if (ctx->dAux) HIP_TRY(ctx, hipMemsetAsync(ctx->dAux, 0, npix*sizeof(TgHipAuxPixel), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

int tghip_bind_framebuffer(tghip_ctx *ctx, float *dev_rgb_sum, uint32_t *dev_count)
{
    if (!ctx) return TGHIP_E_INVALID;
    if ((dev_rgb_sum == nullptr) != (dev_count == nullptr)) { ctx->error = "bind both buffers or neither"; return TGHIP_E_INVALID; }
    ctx->extSum = dev_rgb_sum;
    ctx->extCount = dev_count;
    return TGHIP_OK;
}

extern "C++" {
template<uint32_t M, int FUSE>
static void launchShadeVariant(tghip_ctx *ctx, int grid, const PathState &st, const PassParams &pp, int cls)
{
    constexpr uint32_t B = M & ~FEAT_QMC;
    hipLaunchKernelGGL((k_shade<M, (B == MASK_SIMPLE ? SIMPLE_WAVES : B == MASK_LEAN ? LEAN_WAVES : B == MASK_COAT ? COAT_WAVES : 2), FUSE>), dim3(grid),
                       dim3(M == BSDF_MASK_ALL ? ctx->thrShadeAll : cls == 0 ? ctx->thrShadeSimple : ctx->thrShadeComplex), 0, ctx->stream, ctx->scene, st, pp, cls);
}
// TGHIP_PASS_SOBOL / TGHIP_PASS_RECORDS passes run the FEAT_QMC twin of the variant the scene would use anyway
template<uint32_t M, int FUSE = 0>
static void launchShade(tghip_ctx *ctx, int grid, const PathState &st, const PassParams &pp, int cls)
{
    if (pp.flags) launchShadeVariant<M | FEAT_QMC, FUSE>(ctx, grid, st, pp, cls);
    else          launchShadeVariant<M, FUSE>(ctx, grid, st, pp, cls);
}

// true when the shadow step needs its second half, k_finish (the dynamic-fetch kernel does not regenerate paths itself)
template<bool COUNT>
static bool launchShadow(tghip_ctx *ctx, int grid, const PathState &st, const PassParams &pp, uint32_t iterTag)
{
    const size_t ldsBytes = traceLdsBytes(ctx, ctx->thrShadow);
    const bool flat = isFlat(ctx);
    const bool closestWalk = ctx->haveForward || ctx->haveMeshLight;   // shadow rays are closest-hit walks, not any-hit queries
    if (ctx->haveInstances) {
#define SHADOW_INST(FWD, I) hipLaunchKernelGGL((k_trace_shadow<COUNT, FWD, false, I>), dim3(grid), dim3(ctx->thrShadow), ldsBytes, ctx->stream, ctx->scene, st, pp, iterTag)
        if (closestWalk) { if (ctx->haveSolids) SHADOW_INST(true, 1); else SHADOW_INST(true, 2); }
        else             { if (ctx->haveSolids) SHADOW_INST(false, 1); else SHADOW_INST(false, 2); }
#undef SHADOW_INST
        return false;
    }
    if (!flat && !closestWalk && ctx->dynamicFetch && !ctx->auxPass) {   // (the dynamic-fetch kernel does not report transmittances)
        if (ctx->haveSolids) hipLaunchKernelGGL((k_trace_shadow_dyn<COUNT, true>), dim3(grid), dim3(ctx->thrShadow), dynLdsBytes(ctx, ctx->thrShadow), ctx->stream,
                                                ctx->scene, st, pp, iterTag);
        else                 hipLaunchKernelGGL((k_trace_shadow_dyn<COUNT, false>), dim3(grid), dim3(ctx->thrShadow), dynLdsBytes(ctx, ctx->thrShadow), ctx->stream,
                                                ctx->scene, st, pp, iterTag);
        return true;
    }
#define SHADOW_LAUNCH(FWD, FLAT) hipLaunchKernelGGL((k_trace_shadow<COUNT, FWD, FLAT>), dim3(grid), dim3(ctx->thrShadow), ldsBytes, ctx->stream, ctx->scene, st, pp, iterTag)
    if (closestWalk) { if (flat) SHADOW_LAUNCH(true, true); else SHADOW_LAUNCH(true, false); }
    else                  { if (flat) SHADOW_LAUNCH(false, true); else SHADOW_LAUNCH(false, false); }
#undef SHADOW_LAUNCH
    return false;
}

} // extern "C++"

// Runs the wavefront loop for one batch of work items until every slot is drained.
static int runBatch(tghip_ctx *ctx, const PassParams &pp)
{
    PathState st = ctx->pool;
    st.partial = ctx->partial;
    st.leaf_batch = uint32_t(ctx->leafBatch);
    const DeviceScene &s = ctx->scene;
    const int grid = int(ctx->poolGrid);
    const bool count = ctx->countTraversal;
    const bool flat = isFlat(ctx);
    const bool fused = flat && !ctx->haveForward && !ctx->haveMeshLight && ctx->fuseFlatOpt && !ctx->auxPass && !ctx->haveCylinder;
    const bool runToCompletion = fused && !ctx->haveComplex && ctx->loopOpt;   // one launch renders the whole batch
    const size_t ldsBytes = traceLdsBytes(ctx, ctx->thrClosest);

    // optional per-launch timing: one event pair per kernel launch of a check interval, read back at the
    // interval's host sync (events live on ctx->stream, the stream the kernels are launched on)
    const bool timing = ctx->timeKernels;
    const size_t evNeeded = size_t(ctx->checkInterval)*3*2;
    if (timing) {
        while (ctx->evPool.size() < evNeeded) {
            hipEvent_t e = nullptr;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->evPool.push_back(e);
        }
    }
    size_t evUsed = 0;
    auto tic = [&]() { if (timing) (void)hipEventRecord(ctx->evPool[evUsed++], ctx->stream); };

    HIP_TRY(ctx, hipMemsetAsync(st.partial, 0, size_t(pp.total_items)*sizeof(float4), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(st.live, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_start, dim3(grid), dim3(256), 0, ctx->stream, s, st, pp);
    uint32_t iterTag = 1;                        // k_start publishes tag 1 when it queued anything
    bool first = true;
    int roundIters = ctx->checkInterval;         // launches of the wavefront loop between two host checks
    for (;;) {
        evUsed = 0;
        {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->hostLive, st.live, 2*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (timing && !first) {
                double *acc[3] = {&ctx->counters.ms_trace_closest, &ctx->counters.ms_shade, &ctx->counters.ms_trace_shadow};
                size_t pairs = size_t(roundIters)*3;
                for (size_t k = 0; k < pairs; ++k) {
                    float ms = 0.0f;
                    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evPool[2*k], ctx->evPool[2*k + 1]));
                    *acc[k % 3] += ms;
                }
                ctx->counters.launches_trace_closest += roundIters;
                ctx->counters.launches_trace_shadow += roundIters;
                ctx->counters.launches_shade += roundIters;
            }
            if (ctx->hostLive[0] != iterTag)
                break;                           // the last iteration left every extension queue empty
        }
        first = false;
        roundIters = runToCompletion ? 1 : ctx->checkInterval;
        for (int it = 0; it < roundIters; ++it) {
            ++iterTag;
            if (fused) {
                // flat-list scene: intersection and shadow tests happen inside the shading launches
                PassParams ppi = pp;
                ppi.iter_tag = iterTag;
                tic(); tic(); tic();
                if (runToCompletion) {
                    if (ctx->leanScene) launchShade<MASK_LEAN, FUSE_TRACE | FUSE_SHADOW | FUSE_LOOP>(ctx, grid, st, ppi, 0);
                    else                launchShade<MASK_SIMPLE, FUSE_TRACE | FUSE_SHADOW | FUSE_LOOP>(ctx, grid, st, ppi, 0);
                }
                else if (ctx->leanScene) launchShade<MASK_LEAN, FUSE_TRACE | FUSE_SHADOW>(ctx, grid, st, ppi, 0);
                else                     launchShade<MASK_SIMPLE, FUSE_TRACE | FUSE_SHADOW>(ctx, grid, st, ppi, 0);
                if (ctx->haveComplex) {
                    if ((ctx->complexMask & ~MASK_COAT) == 0)       launchShade<MASK_COAT, FUSE_SHADOW>(ctx, grid, st, ppi, 1);
                    else if ((ctx->complexMask & ~MASK_GLASS) == 0) launchShade<MASK_GLASS, FUSE_SHADOW>(ctx, grid, st, ppi, 1);
                    else                                            launchShade<MASK_FULL, FUSE_SHADOW>(ctx, grid, st, ppi, 1);
                }
                tic(); tic(); tic();
                ctx->counters.iterations++;
                continue;
            }
            tic();
            if (flat) {
                if (count) hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->stream, s, st);
                else       hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->stream, s, st);
            } else if (ctx->haveInstances) {
#define CLOSEST_INST(C, I) hipLaunchKernelGGL((k_trace_closest<C, false, I>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->stream, s, st)
                if (ctx->haveSolids) { if (count) CLOSEST_INST(true, 1); else CLOSEST_INST(false, 1); }
                else                 { if (count) CLOSEST_INST(true, 2); else CLOSEST_INST(false, 2); }
#undef CLOSEST_INST
            } else {
                if (ctx->dynamicFetch) {
                    const size_t ldsDyn = dynLdsBytes(ctx, ctx->thrClosest);
#define CLOSEST_DYN(C, S) hipLaunchKernelGGL((k_trace_closest_dyn<C, S>), dim3(grid), dim3(ctx->thrClosest), ldsDyn, ctx->stream, s, st)
                    if (ctx->haveSolids) { if (count) CLOSEST_DYN(true, true); else CLOSEST_DYN(false, true); }
                    else                 { if (count) CLOSEST_DYN(true, false); else CLOSEST_DYN(false, false); }
#undef CLOSEST_DYN
                } else {
                    if (count) hipLaunchKernelGGL((k_trace_closest<true, false>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->stream, s, st);
                    else       hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->stream, s, st);
                }
            }
            tic(); tic();
            if (ctx->haveMedia || ctx->auxPass || ctx->haveCylinder) {   // the one variant with FEAT_MEDIA / FEAT_AUX / FEAT_CYLINDER, for both classes
                launchShadeVariant<BSDF_MASK_ALL, 0>(ctx, grid, st, pp, 0);
                if (ctx->haveComplex) launchShadeVariant<BSDF_MASK_ALL, 0>(ctx, grid, st, pp, 1);
            }
            else if (ctx->haveMeshLight || ctx->haveInstances) launchShade<MASK_FULL>(ctx, grid, st, pp, 0);   // the only variants with mesh-emitter sampling / instance transforms
            else if (ctx->leanScene) launchShade<MASK_LEAN>(ctx, grid, st, pp, 0);
            else                     launchShade<MASK_SIMPLE>(ctx, grid, st, pp, 0);
            if (ctx->haveComplex && !ctx->haveMedia && !ctx->auxPass && !ctx->haveCylinder) {
                if (ctx->haveMeshLight || ctx->haveInstances)   launchShade<MASK_FULL>(ctx, grid, st, pp, 1);
                else if ((ctx->complexMask & ~MASK_COAT) == 0)  launchShade<MASK_COAT>(ctx, grid, st, pp, 1);
                else if ((ctx->complexMask & ~MASK_GLASS) == 0) launchShade<MASK_GLASS>(ctx, grid, st, pp, 1);
                else                                            launchShade<MASK_FULL>(ctx, grid, st, pp, 1);
            }
            tic(); tic();
            const bool finish = count ? launchShadow<true>(ctx, grid, st, pp, iterTag) : launchShadow<false>(ctx, grid, st, pp, iterTag);
            tic();
            if (finish)
                hipLaunchKernelGGL(k_finish, dim3(grid), dim3(256), 0, ctx->stream, s, st, pp, iterTag);
            ctx->counters.iterations++;
        }
    }
    float *sum = ctx->extSum ? ctx->extSum : ctx->fbSum;
    uint32_t *cnt = ctx->extCount ? ctx->extCount : ctx->fbCount;
    if (pp.rec_sorted) hipLaunchKernelGGL(k_resolve_records, dim3((pp.num_sorted*16u + 255)/256), dim3(256), 0, ctx->stream, st, pp, sum, cnt);
    else               hipLaunchKernelGGL(k_resolve, dim3((pp.pix_slots + 255)/256), dim3(256), 0, ctx->stream, st, pp, sum, cnt);
    HIP_TRY(ctx, hipGetLastError());
    return ctx->hostLive[1] ? TGHIP_E_ABORTED : TGHIP_OK;
}

// The pass itself is driven synchronously from tghip_wait (the integrator calls it from its worker
// thread, which gives the reference's "startRender returns immediately" contract).
int tghip_render_pass(tghip_ctx *ctx, const TgHipPassDesc *pass)
{
    if (!ctx || !pass) return TGHIP_E_INVALID;
    if (!ctx->haveScene) { ctx->error = "render before upload"; return TGHIP_E_NOSCENE; }
    if (pass->spp_end < pass->spp_begin || (pass->shard_count && pass->shard_index >= pass->shard_count)) {
        ctx->error = "invalid pass description";
        return TGHIP_E_INVALID;
    }
    if (pass->flags & ~(TGHIP_PASS_SOBOL | TGHIP_PASS_RECORDS | TGHIP_PASS_AUX)) { ctx->error = "unknown pass flags"; return TGHIP_E_INVALID; }
    if ((pass->flags & TGHIP_PASS_SOBOL) && (!ctx->scene.sobol || !pass->tile_seeds)) {
        ctx->error = "TGHIP_PASS_SOBOL needs sobol_matrices in the scene description and tile_seeds in the pass";
        return TGHIP_E_INVALID;
    }
    if ((pass->record_count != nullptr) != (pass->record_index != nullptr) || (pass->record_count && !(pass->flags & TGHIP_PASS_RECORDS))) {
        ctx->error = "record_index and record_count go together and need TGHIP_PASS_RECORDS";
        return TGHIP_E_INVALID;
    }
    ctx->passPending = true;
    ctx->passResult = TGHIP_OK;
    ctx->pendingPass = *pass;
    return TGHIP_OK;
}

int tghip_wait(tghip_ctx *ctx)
{
    if (!ctx) return TGHIP_E_INVALID;
    if (!ctx->passPending)
        return ctx->passResult;
    ctx->passPending = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const TgHipPassDesc pass = ctx->pendingPass;
    const uint32_t shardCount = pass.shard_count ? pass.shard_count : 1;
    const uint32_t w = ctx->width, h = ctx->height;
    const uint32_t tilesX = (w + 15)/16, tilesY = (h + 15)/16;
    const uint32_t numTiles = tilesX*tilesY;
    const uint32_t ownedTiles = numTiles > pass.shard_index ? (numTiles - pass.shard_index + shardCount - 1)/shardCount : 0;
    uint32_t spp = pass.spp_end - pass.spp_begin, sppBegin = pass.spp_begin;
    PassParams base{};
    base.flags = pass.flags | (ctx->thinlens ? PT_PASS_THINLENS : 0u) | (ctx->haveMedia ? PT_PASS_MEDIA : 0u);
    base.variance_w = (w + 3)/4;
    if (pass.flags & TGHIP_PASS_SOBOL) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dTileSeeds, pass.tile_seeds, size_t(numTiles)*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        base.tile_seeds = ctx->dTileSeeds;
    }
    const uint32_t numRecords = base.variance_w*((h + 3)/4);
    if (pass.flags & TGHIP_PASS_RECORDS) {
        // per record: first sample index, samples per pixel, offset of its 16 x count luminance block.  The sample
        // range of the batches below becomes relative to each record's first index: [0, max count of an owned record).
        std::vector<uint32_t> &idx = ctx->hostRecIndex, &cnt = ctx->hostRecCount, &off = ctx->hostRecLum;
        idx.resize(numRecords); cnt.resize(numRecords); off.resize(numRecords);
        uint64_t total = 0;
        uint32_t maxCount = 0;
        for (uint32_t r = 0; r < numRecords; ++r) {
            uint32_t rx = r % base.variance_w, ry = r/base.variance_w;
            bool owned = ((rx >> 2) + (ry >> 2)*tilesX) % shardCount == pass.shard_index;
            idx[r] = pass.record_index ? pass.record_index[r] : pass.spp_begin;
            cnt[r] = pass.record_count ? pass.record_count[r] : spp;
            off[r] = uint32_t(total);
            if (owned) {
                total += uint64_t(cnt[r])*16u;
                maxCount = std::max(maxCount, cnt[r]);
            }
            if (total >= (1ull << 32)) {
                ctx->error = "pass too large for SampleRecord keeping (more than 2^32 samples per device): lower spp_step";
                return ctx->passResult = TGHIP_E_UNSUPPORTED;
            }
        }
        spp = maxCount;
        sppBegin = 0;
        if (ctx->lumCap < total) {
            if (ctx->lum) (void)hipFree(ctx->lum);
            ctx->lum = nullptr; ctx->lumCap = 0;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->lum), std::max<uint64_t>(total, 1)*sizeof(float)));
            ctx->lumCap = total;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dRecIndex, idx.data(), numRecords*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dRecCount, cnt.data(), numRecords*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dRecLum, off.data(), numRecords*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        base.rec_index = ctx->dRecIndex; base.rec_count = ctx->dRecCount; base.rec_lum = ctx->dRecLum;
        base.lum = ctx->lum;
    }
    const uint32_t sppEnd = sppBegin + spp;
    if (ownedTiles == 0 || spp == 0)
        return ctx->passResult = TGHIP_OK;

    // TGHIP_PASS_AUX: one work item per pixel, so that a pixel's samples reach OutputBuffer::addSample in index order (auxAdd)
    ctx->auxPass = (pass.flags & TGHIP_PASS_AUX) != 0;
    const uint32_t chunk = ctx->auxPass ? std::max(spp, 1u) : uint32_t(std::max(ctx->chunkSamples, 1));
    if (ctx->auxPass && !ctx->dAux) {
        const size_t npix = size_t(w)*h;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dAux), npix*sizeof(TgHipAuxPixel)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dAux, 0, npix*sizeof(TgHipAuxPixel), ctx->stream));
    }
    base.aux = ctx->dAux;
    const uint64_t maxItems = uint64_t(std::max<long long>(ctx->maxItems, 256));
    const bool recordPass = (pass.flags & TGHIP_PASS_RECORDS) != 0;
    uint64_t recordItems = 0;
    if (recordPass) {
        // Gap-free work items for uneven per-record sample counts (PassParams): owned records sorted by descending count,
        // chunk c covers the first chunkStart[c + 1] - chunkStart[c] pixel slots of that order.
        std::vector<uint32_t> &sorted = ctx->hostSorted, &start = ctx->hostChunkStart, &hint = ctx->hostHint;
        const std::vector<uint32_t> &cnt = ctx->hostRecCount;
        sorted.clear();
        for (uint32_t r = 0; r < numRecords; ++r) {
            uint32_t rx = r % base.variance_w, ry = r/base.variance_w;
            if (((rx >> 2) + (ry >> 2)*tilesX) % shardCount == pass.shard_index && cnt[r] > 0)
                sorted.push_back(r);
        }
        // descending by count, ties in record order (a counting sort: the counts of a pass are small integers)
        if (spp < (1u << 16)) {
            std::vector<uint32_t> &bucket = ctx->hostBucket;
            bucket.assign(size_t(spp) + 2, 0u);
            for (uint32_t r : sorted) bucket[spp - cnt[r] + 1]++;
            for (size_t i = 1; i < bucket.size(); ++i) bucket[i] += bucket[i - 1];
            std::vector<uint32_t> &tmp = ctx->hostSortTmp;
            tmp.resize(sorted.size());
            for (uint32_t r : sorted) tmp[bucket[spp - cnt[r]]++] = r;
            sorted.swap(tmp);
        } else {
            std::stable_sort(sorted.begin(), sorted.end(), [&cnt](uint32_t a, uint32_t b) { return cnt[a] > cnt[b]; });
        }
        const uint32_t numChunks = (spp + chunk - 1)/chunk;
        start.assign(size_t(numChunks) + 2, 0u);
        size_t alive = sorted.size();                // records with cnt > c*chunk: a prefix of `sorted`
        for (uint32_t c = 0; c < numChunks; ++c) {
            while (alive > 0 && cnt[sorted[alive - 1]] <= uint64_t(c)*chunk) --alive;
            recordItems += uint64_t(alive)*16u;
            if (recordItems >= (1ull << 32)) {
                ctx->error = "pass too large for SampleRecord keeping (more than 2^32 work items per device): lower spp_step";
                return ctx->passResult = TGHIP_E_UNSUPPORTED;
            }
            start[c + 1] = uint32_t(recordItems);
        }
        start[numChunks + 1] = 0xFFFFFFFFu;          // sentinel for the device's `while (w >= start[c + 1])`
        hint.resize(size_t((recordItems + 63)/64) + 1);
        for (size_t g = 0, c = 0; g < hint.size(); ++g) {
            while (c + 1 < numChunks && uint64_t(g)*64 >= start[c + 1]) ++c;
            hint[g] = uint32_t(c);
        }
        auto upload = [&](uint32_t *&dev, size_t &cap, const std::vector<uint32_t> &src) -> int {
            if (cap < src.size()) {
                if (dev) (void)hipFree(dev);
                dev = nullptr; cap = 0;
                HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&dev), std::max<size_t>(src.size(), 1)*sizeof(uint32_t)));
                cap = src.size();
            }
            if (!src.empty())
                HIP_TRY(ctx, hipMemcpyAsync(dev, src.data(), src.size()*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            return TGHIP_OK;
        };
        int urc;
        if ((urc = upload(ctx->dSorted, ctx->sortedCap, sorted)) != TGHIP_OK) return ctx->passResult = urc;
        if ((urc = upload(ctx->dChunkStart, ctx->chunkStartCap, start)) != TGHIP_OK) return ctx->passResult = urc;
        if ((urc = upload(ctx->dHint, ctx->hintCap, hint)) != TGHIP_OK) return ctx->passResult = urc;
        base.rec_sorted = ctx->dSorted; base.rec_chunk_start = ctx->dChunkStart; base.rec_hint = ctx->dHint;
        base.num_sorted = uint32_t(sorted.size());
        base.num_chunks = numChunks;
        if (recordItems == 0)
            return ctx->passResult = TGHIP_OK;
    }

    // Batches: work items = (pixel slot of an owned tile) x (chunk of `chunkSamples` sample indices).  One batch
    // holds at most maxItems items (16 B of partial sum each); larger passes are split by sample range first,
    // then by tiles.  (Record passes: consecutive ranges of the gap-free enumeration above.)
    const uint32_t chunksAll = (spp + chunk - 1)/chunk;
    uint32_t tilesPerBatch = ownedTiles, chunksPerBatch = chunksAll;
    if (!recordPass && uint64_t(ownedTiles)*256*chunksAll > maxItems) {
        chunksPerBatch = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(chunksAll, maxItems/(uint64_t(ownedTiles)*256))));
        if (uint64_t(ownedTiles)*256*chunksPerBatch > maxItems)
            tilesPerBatch = uint32_t(std::max<uint64_t>(1, maxItems/(256ull*chunksPerBatch)));
    }
    const uint64_t batchItems = recordPass ? std::min<uint64_t>(recordItems, maxItems) : uint64_t(tilesPerBatch)*256*chunksPerBatch;
    uint64_t wantSlots = std::min<uint64_t>(uint64_t(ctx->maxSlots), batchItems);
    {
        // the run-to-completion kernel (runBatch) gives every thread its own slots for the whole launch: one slot per
        // thread keeps the whole path state (112 B x 0.5 M slots) inside the Infinity Cache -- measured +5 % over four
        const bool flat = isFlat(ctx);
        const bool loop = flat && !ctx->haveForward && !ctx->haveMeshLight && ctx->fuseFlatOpt && !ctx->haveComplex && ctx->loopOpt && !ctx->auxPass && !ctx->haveCylinder;
        if (loop && !ctx->maxSlotsSet)
            wantSlots = std::min<uint64_t>(wantSlots, uint64_t(launchGrid(ctx))*uint64_t(ctx->thrShadeSimple));
    }
    const uint32_t slots = uint32_t(wantSlots);
    int rc = ensurePool(ctx, slots);
    if (rc != TGHIP_OK) return ctx->passResult = rc;
    if (ctx->partialCap < batchItems) {
        if (ctx->partial) (void)hipFree(ctx->partial);
        ctx->partial = nullptr; ctx->partialCap = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->partial), batchItems*sizeof(float4)));
        ctx->partialCap = batchItems;
    }
    HIP_TRY(ctx, hipMemsetAsync(&ctx->pool.live[1], 0, sizeof(uint32_t), ctx->stream));   // abort flag

    HIP_TRY(ctx, hipEventRecord(ctx->evA, ctx->stream));
    for (uint64_t w0 = 0; recordPass && w0 < recordItems && rc == TGHIP_OK; w0 += batchItems) {
        PassParams pp = base;
        pp.spp_begin = 0; pp.spp_end = spp; pp.seed = pass.seed;
        pp.chunk = chunk;
        pp.chunks = chunksAll;
        pp.pix_slots = 1;                            // unused by the record enumeration
        pp.item_base = uint32_t(w0);
        pp.total_items = uint32_t(std::min<uint64_t>(batchItems, recordItems - w0));
        pp.first_tile = 0;
        pp.shard_index = pass.shard_index; pp.shard_count = shardCount;
        pp.tiles_x = tilesX; pp.num_tiles = numTiles;
        pp.width = w; pp.height = h;
        rc = runBatch(ctx, pp);
    }
    for (uint32_t sppFirst = sppBegin; !recordPass && sppFirst < sppEnd && rc == TGHIP_OK; sppFirst += chunksPerBatch*chunk) {
        const uint32_t sppLast = uint32_t(std::min<uint64_t>(uint64_t(sppFirst) + uint64_t(chunksPerBatch)*chunk, sppEnd));
        for (uint32_t first = 0; first < ownedTiles; first += tilesPerBatch) {
            PassParams pp = base;
            pp.spp_begin = sppFirst; pp.spp_end = sppLast; pp.seed = pass.seed;
            pp.chunk = chunk;
            pp.chunks = (sppLast - sppFirst + chunk - 1)/chunk;
            pp.pix_slots = std::min(tilesPerBatch, ownedTiles - first)*256;
            pp.total_items = pp.pix_slots*pp.chunks;
            pp.first_tile = first;
            pp.shard_index = pass.shard_index; pp.shard_count = shardCount;
            pp.tiles_x = tilesX; pp.num_tiles = numTiles;
            pp.width = w; pp.height = h;
            rc = runBatch(ctx, pp);
            if (rc != TGHIP_OK) break;
        }
    }
    if ((pass.flags & TGHIP_PASS_RECORDS) && rc == TGHIP_OK) {
        PassParams pp = base;
        pp.shard_index = pass.shard_index; pp.shard_count = shardCount;
        pp.tiles_x = tilesX; pp.num_tiles = numTiles;
        pp.width = w; pp.height = h;
        hipLaunchKernelGGL(k_records, dim3((numRecords + 63)/64), dim3(64), 0, ctx->stream, pp, ctx->dRecords, numRecords);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(ctx->evB, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evA, ctx->evB));
    ctx->counters.ms_total += ms;
    return ctx->passResult = rc;
}

int tghip_abort(tghip_ctx *ctx)
{
    if (!ctx) return TGHIP_E_INVALID;
    if (!ctx->poolSlots) return TGHIP_OK;
    // device-visible flag polled whenever a slot asks for new work; written from a second stream so it
    // does not queue behind the running pass
    static const uint32_t one = 1;
    hipStream_t side = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipError_t e = hipMemcpyAsync(&ctx->pool.live[1], &one, sizeof(one), hipMemcpyHostToDevice, side);
    if (e == hipSuccess) e = hipStreamSynchronize(side);
    (void)hipStreamDestroy(side);
    if (e != hipSuccess) { ctx->error = hipGetErrorString(e); return TGHIP_E_HIP; }
    return TGHIP_OK;
}

int tghip_download_framebuffer(tghip_ctx *ctx, float *rgb_sum, uint32_t *count, size_t npixels)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    if (npixels != size_t(ctx->width)*ctx->height) { ctx->error = "pixel count mismatch"; return TGHIP_E_INVALID; }
    int rc = tghip_wait(ctx);
    if (rc != TGHIP_OK && rc != TGHIP_E_ABORTED) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const float *sum = ctx->extSum ? ctx->extSum : ctx->fbSum;
    const uint32_t *cnt = ctx->extCount ? ctx->extCount : ctx->fbCount;
    if (rgb_sum) HIP_TRY(ctx, hipMemcpyAsync(rgb_sum, sum, npixels*3*sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (count) HIP_TRY(ctx, hipMemcpyAsync(count, cnt, npixels*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

int tghip_upload_framebuffer(tghip_ctx *ctx, const float *rgb_sum, const uint32_t *count, size_t npixels)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    if (!rgb_sum || !count || npixels != size_t(ctx->width)*ctx->height) { ctx->error = "pixel count mismatch"; return TGHIP_E_INVALID; }
    int rc = tghip_wait(ctx);
    if (rc != TGHIP_OK && rc != TGHIP_E_ABORTED) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *sum = ctx->extSum ? ctx->extSum : ctx->fbSum;
    uint32_t *cnt = ctx->extCount ? ctx->extCount : ctx->fbCount;
    HIP_TRY(ctx, hipMemcpyAsync(sum, rgb_sum, npixels*3*sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(cnt, count, npixels*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

int tghip_download_records(tghip_ctx *ctx, TgHipSampleRecord *out, size_t n)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    if (!out || n != size_t((ctx->width + 3)/4)*((ctx->height + 3)/4)) { ctx->error = "record count mismatch"; return TGHIP_E_INVALID; }
    int rc = tghip_wait(ctx);
    if (rc != TGHIP_OK && rc != TGHIP_E_ABORTED) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->dRecords, n*sizeof(TgHipSampleRecord), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

int tghip_upload_records(tghip_ctx *ctx, const TgHipSampleRecord *in, size_t n)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    if (!in || n != size_t((ctx->width + 3)/4)*((ctx->height + 3)/4)) { ctx->error = "record count mismatch"; return TGHIP_E_INVALID; }
    int rc = tghip_wait(ctx);
    if (rc != TGHIP_OK && rc != TGHIP_E_ABORTED) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dRecords, in, n*sizeof(TgHipSampleRecord), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

int tghip_download_aux(tghip_ctx *ctx, TgHipAuxPixel *out, size_t npixels)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    if (!out || npixels != size_t(ctx->width)*ctx->height) { ctx->error = "pixel count mismatch"; return TGHIP_E_INVALID; }
    int rc = tghip_wait(ctx);
    if (rc != TGHIP_OK && rc != TGHIP_E_ABORTED) return rc;
    if (!ctx->dAux) { std::memset(out, 0, npixels*sizeof(TgHipAuxPixel)); return TGHIP_OK; }   // no TGHIP_PASS_AUX pass yet
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->dAux, npixels*sizeof(TgHipAuxPixel), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

int tghip_upload_aux(tghip_ctx *ctx, const TgHipAuxPixel *in, size_t npixels)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    if (!in || npixels != size_t(ctx->width)*ctx->height) { ctx->error = "pixel count mismatch"; return TGHIP_E_INVALID; }
    int rc = tghip_wait(ctx);
    if (rc != TGHIP_OK && rc != TGHIP_E_ABORTED) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->dAux) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dAux), npixels*sizeof(TgHipAuxPixel)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dAux, in, npixels*sizeof(TgHipAuxPixel), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

int tghip_trace_rays(tghip_ctx *ctx, const TgHipRay *rays, TgHipHit *hits, size_t n, int repeats, double *ms_per_launch)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    if (n == 0) return TGHIP_OK;
    if (!rays || !hits || n > 0x7FFFFFFFu) { ctx->error = "invalid ray batch"; return TGHIP_E_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float4 *dRays = nullptr, *dHits = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&dRays), n*sizeof(TgHipRay)));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dHits), n*sizeof(TgHipHit));
    if (e != hipSuccess) { (void)hipFree(dRays); ctx->error = hipGetErrorString(e); return TGHIP_E_HIP; }
    int rc = ensurePool(ctx, ctx->poolSlots ? ctx->pool.num_slots : 256);   // the traversal statistics live in the pool
    if (rc != TGHIP_OK) { (void)hipFree(dRays); (void)hipFree(dHits); return rc; }
    (void)hipMemcpyAsync(dRays, rays, n*sizeof(TgHipRay), hipMemcpyHostToDevice, ctx->stream);
    const int grid = int(std::min<size_t>(size_t(launchGrid(ctx)), (n + 255)/256));
    const size_t ldsBytes = traceLdsBytes(ctx, 256);
    repeats = std::max(repeats, 1);
    (void)hipEventRecord(ctx->evA, ctx->stream);
    for (int r = 0; r < repeats; ++r) {
        const bool cnt = ctx->countTraversal && r == 0;
        const bool flat = isFlat(ctx);
#define RAYS_LAUNCH(C, F) hipLaunchKernelGGL((k_trace_rays<C, F>), dim3(grid), dim3(256), ldsBytes, ctx->stream, ctx->scene, dRays, dHits, uint32_t(n), ctx->pool.stats)
        if (ctx->haveInstances) {
            if (cnt) hipLaunchKernelGGL((k_trace_rays<true, false, 1>), dim3(grid), dim3(256), ldsBytes, ctx->stream, ctx->scene, dRays, dHits, uint32_t(n), ctx->pool.stats);
            else     hipLaunchKernelGGL((k_trace_rays<false, false, 1>), dim3(grid), dim3(256), ldsBytes, ctx->stream, ctx->scene, dRays, dHits, uint32_t(n), ctx->pool.stats);
        }
        else if (cnt) { if (flat) RAYS_LAUNCH(true, true); else RAYS_LAUNCH(true, false); }
        else          { if (flat) RAYS_LAUNCH(false, true); else RAYS_LAUNCH(false, false); }
#undef RAYS_LAUNCH
    }
    (void)hipEventRecord(ctx->evB, ctx->stream);
    (void)hipMemcpyAsync(hits, dHits, n*sizeof(TgHipHit), hipMemcpyDeviceToHost, ctx->stream);
    e = hipStreamSynchronize(ctx->stream);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->evA, ctx->evB);
    (void)hipFree(dRays); (void)hipFree(dHits);
    if (e != hipSuccess) { ctx->error = hipGetErrorString(e); return TGHIP_E_HIP; }
    if (ms_per_launch) *ms_per_launch = double(ms)/repeats;
    return TGHIP_OK;
}

int tghip_get_counters(tghip_ctx *ctx, TgHipCounters *out)
{
    if (!ctx || !out) return TGHIP_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = foldCounters(ctx);
    if (rc != TGHIP_OK) return rc;
    *out = ctx->counters;
    return TGHIP_OK;
}

int tghip_reset_counters(tghip_ctx *ctx)
{
    if (!ctx) return TGHIP_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = foldCounters(ctx);
    if (rc != TGHIP_OK) return rc;
    std::memset(&ctx->counters, 0, sizeof(ctx->counters));
    return TGHIP_OK;
}

} // extern "C"
