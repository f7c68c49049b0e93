// Wavefront kernels of path_tracer_hip (gfx950): every __global__ template of the tracer.  Included by tungsten_hip.hip (the
// extern "C" shim, which launches them) and by the shade_*.hip translation units, which hold the explicit instantiations of
// the k_shade variants so that the ~70 kernel instantiations compile in parallel.  See pt_kernels.h for the execution model.
#ifndef TGAMD_PT_WAVEFRONT_H_
#define TGAMD_PT_WAVEFRONT_H_
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
#define PROF_DECL unsigned long long profT = wall_clock64(), profAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, profLn[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
/* (round 6: profLn = the section's ticks times the lanes enabled at its end marker -- profLn / profAcc is the section's lane count as lane 0 sees it) */
#define PROF(n) do { unsigned long long t_ = wall_clock64(); profAcc[n] += t_ - profT; profLn[n] += (t_ - profT)*(unsigned long long)__popcll(__ballot(true)); profT = t_; if ((n) == 0) profAcc[10]++; } while (0)   /* ticks of 10 ns; [10] = turns */
#define PROF_FLUSH(stats) do { if (laneId() == 0) for (int k_ = 0; k_ < 16; ++k_) if (k_ != 11) { if (k_ != 10) atomicAdd(&(stats).prof[k_], profAcc[k_]); atomicAdd(&(stats).profCls[cls][k_], profAcc[k_]); atomicAdd(&(stats).profLanes[cls][k_], profLn[k_]); } } while (0)
#else
#define PROF_DECL
#define PROF(n)
#define PROF_FLUSH(stats)
#endif

// count_traversal: time and lane-occupancy breakdown of the wide traversal kernels (BlockStats::walk)
#define WALK_PROF_DECL unsigned long long wpLoop = COUNT ? wall_clock64() : 0ull, wpDry = 0ull, wpLoopEnd = 0ull; uint32_t wpBusy = 0, wpBusyDry = 0, wpSuspended = 0, wpResumed = 0, \
    wpRecTurns = 0, wpRecLanes = 0, wpNodeTurns = 0, wpNodeLanes = 0, wpRefills = 0, wpRefillLanes = 0, wpPubs = 0, wpPubLanes = 0, wpHits = 0, wpRays = 0
// (wave-uniform tallies of a section: how often the wave ran it, and how many lanes had work in it)
#define WALK_SECTION(TURNS, LANES, pred) do { if (COUNT) { const unsigned long long m_ = __ballot(pred); if (m_) { TURNS++; LANES += (uint32_t)__popcll(m_); } } } while (0)
#define WALK_PROF_FLUSH(K, TURNS, DRYTURNS) do { \
        const unsigned long long wpEnd_ = wall_clock64(); \
        if (wpDry == 0ull) wpDry = wpLoopEnd; \
        uint32_t su_ = wpSuspended, re_ = wpResumed, hi_ = wpHits, ra_ = wpRays; \
        for (int off_ = 32; off_ > 0; off_ >>= 1) { su_ += __shfl_down(su_, off_); re_ += __shfl_down(re_, off_); hi_ += __shfl_down(hi_, off_); ra_ += __shfl_down(ra_, off_); } \
        if (laneId() == 0) { \
            unsigned long long *wk_ = st.stats[blockIdx.x].walk[K]; \
            atomicAdd(&wk_[0], wpLoop - wpStart); atomicAdd(&wk_[1], wpDry - wpLoop); atomicAdd(&wk_[2], wpLoopEnd - wpDry); atomicAdd(&wk_[3], wpEnd_ - wpLoopEnd); \
            atomicAdd(&wk_[4], 1ull); atomicAdd(&wk_[5], (unsigned long long)(TURNS)); atomicAdd(&wk_[6], (unsigned long long)(DRYTURNS)); \
            atomicAdd(&wk_[7], (unsigned long long)wpBusy); atomicAdd(&wk_[8], (unsigned long long)wpBusyDry); \
            atomicAdd(&wk_[9], (unsigned long long)su_); atomicAdd(&wk_[10], (unsigned long long)re_); atomicMax(&wk_[11], wpLoopEnd - wpLoop); \
            atomicAdd(&wk_[12], (unsigned long long)wpRecTurns); atomicAdd(&wk_[13], (unsigned long long)wpRecLanes); \
            atomicAdd(&wk_[14], (unsigned long long)wpNodeTurns); atomicAdd(&wk_[15], (unsigned long long)wpNodeLanes); \
            atomicAdd(&wk_[16], (unsigned long long)wpRefills); atomicAdd(&wk_[17], (unsigned long long)wpRefillLanes); \
            atomicAdd(&wk_[18], (unsigned long long)wpPubs); atomicAdd(&wk_[19], (unsigned long long)wpPubLanes); \
            atomicAdd(&wk_[20], (unsigned long long)hi_); atomicAdd(&wk_[21], (unsigned long long)ra_); \
        } } while (0)

// BSDF type sets of the shading-kernel variants (pt_scene.h BsdfOps<D, M>)
#define TYPES_SIMPLE (BSDF_BIT(TGHIP_BSDF_LAMBERT) | BSDF_BIT(TGHIP_BSDF_NULL) | BSDF_BIT(TGHIP_BSDF_ERROR))
#define MASK_SIMPLE  (TYPES_SIMPLE | FEAT_ALL)
#define MASK_LEAN    TYPES_SIMPLE        /* analytic primitives, constant/checker textures, one area light (Cornell box) */
#define MASK_SIMPLE_INST (MASK_SIMPLE | FEAT_INSTANCES)   /* classes 0 and 2 of scenes with instance records (no mesh emitters) */
#ifndef SIMPLE_WAVES
#define SIMPLE_WAVES 3   /* measured: 4 waves/SIMD (128 VGPRs, spills) is 10 % slower on materialtest's k_shade */
#endif
#ifndef COAT_WAVES
#define COAT_WAVES   3   /* round 4, without Phong's pow(double) in the family variants (197 VGPRs at 2 waves/SIMD; 168 + 72 B of scratch at 3):
                            materialtest 968.3 / 969.4 -> 977.8 / 974.5, mesh1m 616.9 / 614.2 -> 622.7 / 621.7 Msamples/s (two libraries alternated
                            twice in one session, profiles/r4_ab_coat_waves.txt); round 3 had measured 3 waves as neutral at 223 VGPRs */
#endif
#ifndef PT_TRACE_AHEAD
#define PT_TRACE_AHEAD 0     /* 1: the one-launch render (FUSE_LOOP) traces a slot's next ray at the end of the turn (shadeBody) -- measured, slower: profiles/r6_ab_trace_ahead.txt */
#endif
#ifndef LEAN_WAVES
#define LEAN_WAVES   2   /* measured: 2 waves/SIMD without scratch beats 3 with 108 B of scratch (kernel is VALU-bound) */
#endif
#define MASK_COAT    (MASK_SIMPLE | BSDF_BIT(TGHIP_BSDF_ROUGH_CONDUCTOR) | BSDF_BIT(TGHIP_BSDF_SMOOTH_COAT) | \
                      BSDF_BIT(TGHIP_BSDF_MIRROR) | BSDF_BIT(TGHIP_BSDF_CONDUCTOR))
#define MASK_GLASS   (MASK_SIMPLE | BSDF_BIT(TGHIP_BSDF_DIELECTRIC) | BSDF_BIT(TGHIP_BSDF_ROUGH_DIELECTRIC) | \
                      BSDF_BIT(TGHIP_BSDF_MIRROR))
#define MASK_PLASTIC (MASK_SIMPLE | BSDF_BIT(TGHIP_BSDF_PLASTIC) | BSDF_BIT(TGHIP_BSDF_ROUGH_PLASTIC))
/* media scenes whose surfaces are Lambert / null / forward / (smooth) dielectric / mirror -- every media scene the reference ships and the
   fog / smoke goldens: 216 VGPRs without scratch where BSDF_MASK_ALL spills 292 registers to 848 B of scratch.  (Always the FEAT_QMC twin:
   media passes carry PT_PASS_MEDIA in their flags.) */
#define MASK_MEDIA   (MASK_SIMPLE | FEAT_MEDIA | FEAT_QMC | BSDF_BIT(TGHIP_BSDF_FORWARD) | BSDF_BIT(TGHIP_BSDF_DIELECTRIC) | BSDF_BIT(TGHIP_BSDF_MIRROR))
/* the five types added last (ABI 9): only the full variants shade them; scenes that use one keep the loop to the end (no k_tail) */
#define TYPES_LATE   (BSDF_BIT(TGHIP_BSDF_DIFFUSE_TRANSMISSION) | BSDF_BIT(TGHIP_BSDF_PHONG) | BSDF_BIT(TGHIP_BSDF_THINSHEET) | \
                      BSDF_BIT(TGHIP_BSDF_OREN_NAYAR) | BSDF_BIT(TGHIP_BSDF_ROUGH_COAT))
#define MASK_TAIL    (MASK_FULL & ~(FEAT_INSTANCES | FEAT_MESHLIGHT | TYPES_LATE))   /* k_tail: the 14 BSDF types of rounds 1-3, single-level scenes without mesh emitters */
/* the class variants of scenes with instance records (no mesh emitters): hits reached through an instance (FEAT_INSTANCES) */
#define MASK_COAT_INST    (MASK_COAT | FEAT_INSTANCES)
#define MASK_GLASS_INST   (MASK_GLASS | FEAT_INSTANCES)
#define MASK_PLASTIC_INST (MASK_PLASTIC | FEAT_INSTANCES)

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
template<bool CONVERGED = true, bool EXT = true, int NTS = PT_NT_TRAV>
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
        uint4 sm = slotU4<NTS>(st, A_SAMP, slot), misc = slotU4<NTS>(st, A_MISC, slot);
        samp = make_uint2(sm.x, sm.y);
        lumBase = EXT ? sm.w : 0u;
        acc = slotF4<NTS>(st, A_ACC, slot);
        pixel = misc.z;
        item = misc.w;
        if (black || isnan(sum3(em)))
            em = splat3(0.0f);
        if (records)   // SampleRecord::addSample(c) input (SampleRecord.hpp:55-58; Vec3f::luminance, math/Vec.hpp:195-199)
            {
                const float l = em.x*0.2126f + em.y*0.7152f + em.z*0.0722f;
                if constexpr ((NTS & 4) != 0) __builtin_nontemporal_store(l, &at32(pp.lum, lumBase + samp.x));
                else at32(pp.lum, lumBase + samp.x) = l;
            }
        if constexpr (EXT) {
            if (pp.flags & TGHIP_PASS_SAMPLES) {   // what traceSample returned for (pixel, sample)
                float *dst = pp.samples + ((size_t)pixel*pp.samples_spp + (samp.x - pp.samples_begin))*3u;
                dst[0] = em.x; dst[1] = em.y; dst[2] = em.z;
            }
        }
        if (!(isinf(em.x) || isinf(em.y) || isinf(em.z))) {
            acc.x += em.x; acc.y += em.y; acc.z += em.z;
            acc.w = __uint_as_float(__float_as_uint(acc.w) + 1u);
        }
        if constexpr (EXT) {
            if (pp.flags & TGHIP_PASS_AUX) {
                // the addSample calls of traceSample (PathTracer.cpp:78-96, 133-140), then the colour (PathTraceIntegrator.cpp:152)
                TgHipAuxPixel &px = pp.aux[pixel];
                float4 a0 = slotF4<NTS>(st, A_AUX0, slot), a1 = slotF4<NTS>(st, A_AUX1, slot);
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
            if constexpr ((NTS & 4) != 0) { const PtF4v t = {acc.x, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(t, reinterpret_cast<PtF4v *>(&at32(st.partial, item))); }
            else at32(st.partial, item) = acc;
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
                uint32_t w = (uint32_t)w64 + pp.item_begin;
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
            slotF4<NTS>(st, A_THR, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(FLAG_MAKE(0, 0, ST_DONE)));
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
            const bool cameraOk = cameraRay<EXT>(cam, lens, px, py, l0, l1, xi0, xi1, o, d, s.dist);
            slotF4<NTS>(st, A_RAY_O, slot) = mk4(o, 1e-4f);                      // Ray ctor default nearT (math/Ray.hpp:24)
            slotF4<NTS>(st, A_RAY_D, slot) = mk4(d, cameraOk ? PT_INF : -1.0f);   // a failed camera sample: the ray can hit nothing ...
            slotU4<NTS>(st, A_MISC, slot) = make_uint4((uint32_t)rng.state, (uint32_t)(rng.state >> 32), pixel, item);
            slotF4<NTS>(st, A_EMI, slot) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            // ... and carries no throughput, so the escaped path adds nothing: a black sample (PathTracer.cpp:27-28)
            const float t0 = cameraOk ? 1.0f : 0.0f;
            uint32_t f0 = FLAG_MAKE(0, 1, ST_ACTIVE);                       // wasSpecular starts true
            if (EXT && (pp.flags & PT_PASS_MEDIA))
                f0 |= FLAG_MEDIUM_BITS(cam.medium, 0);                      // _scene->cam().medium(), state.reset() (PathTracer.cpp:38-41)
            slotF4<NTS>(st, A_THR, slot) = make_float4(t0, t0, t0, __uint_as_float(f0));
            if (EXT && (pp.flags & TGHIP_PASS_AUX)) {                          // nothing recorded yet, hitDistance = 0
                const float nan = __uint_as_float(0x7FC00000u);
                slotF4<NTS>(st, A_AUX0, slot) = make_float4(nan, nan, nan, 0.0f);
                slotF4<NTS>(st, A_AUX1, slot) = make_float4(nan, nan, nan, nan);
            }
            slotU4<NTS>(st, A_SAMP, slot) = make_uint4(samp.x, samp.y, EXT ? rng.dim : 0u, lumBase);   // .z: next Sobol' dimension
            slotF4<NTS>(st, A_ACC, slot) = acc;
            push = true;
        }
    }
    return push;
}

// Pass start: every slot takes its first work item.
#ifdef PT_WAVEFRONT_MAIN   /* non-template kernels live in the shim's translation unit only */
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
    uint32_t liveSlots = 0;                      // (BlockCtl::live_slots, as k_finish leaves it)
    for (uint32_t wd = threadIdx.x; wd < (st.slots_per_block >> 5); wd += blockDim.x)
        liveSlots += (uint32_t)__popc(L.bm[Q_EXTP][wd]);
    waveAddStat(&L.nodes, liveSlots);
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl.item_cursor = L.cursor;
        ctl.live_slots = L.nodes;
        if (any) atomicMax(&st.live[0], 1u);
    }
}
#endif

// INST: the scene has instance records (two-level traversal; the instance of a hit goes to the spare word A_EMI.w)
// INST: 0 = single-level scene, 1 = instance records + every record kind, 2 = instance records in a scene of triangles and quads only
template<bool COUNT, bool FLAT, int INST = 0>
__global__ __launch_bounds__(512) void k_trace_closest(DeviceScene s, PathState st)
{
    extern __shared__ int ldsStack[];
    __shared__ BlockLdsSmall L;
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
                slotW(st, A_EMI, slot, 3u) = __int_as_float(hitInst);
            } else {
                hit = traverseClosest<COUNT, FLAT>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims);
            }
            slotF4(st, A_HIT, slot) = hit;
            int ri = __float_as_int(hit.w);
            cls = ri < 0 ? CLS_MISS : (int)at32(s.rec_class, (uint32_t)ri);
            rays++;
        }
        // sort by material: one shading queue per class
        queuePush(cls >= 0, local, L, shadeQueue(cls));
    }
    waveAddStat(&L.closest_rays, rays);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }
    queuesEnd(L, st, Q_EXT, Q_SHADE_MASK, Q_EXTP);
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
// (Scenes with `instances` primitives run k_trace_closest<., ., INST>: the reference's visiting order decides what such a ray hits,
// pt_kernels.h: instanceSetIntersect.)
template<bool COUNT, bool SOLIDS = true>
__global__ __launch_bounds__(512) void k_trace_closest_dyn(DeviceScene s, PathState st)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLdsSmall L;
    __shared__ uint32_t fetchNext;
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    int *stack = ldsDyn + (st.slots_per_block >> 1) + threadIdx.x;
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
            float4 n0 = ld4(nd, 0u), n1 = ld4(nd, 1u), n2 = ld4(nd, 2u), n3 = ld4(nd, 3u);
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
            if (atLeaf != 0ull && ((uint32_t)__popcll(atLeaf) >= st.leaf_batch_bvh2 || atNode == 0ull)) {
                if (busy && cur < 0 && !pop) {
                    uint32_t firstRec = TGHIP_LEAF_FIRST(cur), count = TGHIP_LEAF_COUNT(cur);
                    for (uint32_t r = firstRec; r < firstRec + count; ++r) {
                        if (COUNT) prims++;
                        uint32_t meta;
                        (void)testRecord<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, r, ray, tmax, hit, meta);
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
                int cls = ri < 0 ? CLS_MISS : (int)at32(s.rec_class, (uint32_t)ri);
                queuePush(true, local, L, shadeQueue(cls));
                busy = false;
            } else {
                sp--;
                cur = stack[sp*stride];
            }
        }
    }
    waveAddStat(&L.closest_rays, rays);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }
    queuesEnd(L, st, Q_EXT, Q_SHADE_MASK, Q_EXTP);
    if (threadIdx.x == 0) {
        ctl.closest_rays += L.closest_rays;
        if (COUNT) { st.stats[blockIdx.x].nodes_visited += L.nodes; st.stats[blockIdx.x].prims_tested += L.prims; }
    }
}

// Closest hits of scenes with `instances` primitives, with dynamic ray fetch: the three-level walk of traverseClosestInst (pt_kernels.h) --
// the scene's BVH2, behind an instance-set record the reference's own tree over the instances in the reference's order, behind an instance
// its master's subtree with farT = infinity -- as ONE state machine per lane, one node or one leaf per loop turn, idle lanes refilled from
// the workgroup's queue as in k_trace_closest_dyn.  `level`: 0 the scene's tree, 1 the instance tree, 2 a master's subtree.  All levels
// share one stack (one word per entry: refLeafEntry says why the instance tree needs no entry distances on it).  Same results as
// k_trace_closest<., ., INST>, hit for hit (tests/test_gpu_parity.py).
template<bool COUNT, bool SOLIDS = true>
__global__ __launch_bounds__(512) void k_trace_closest_inst(DeviceScene s, PathState st)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLdsSmall L;
    __shared__ uint32_t fetchNext;
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    int *stack = ldsDyn + (st.slots_per_block >> 1) + threadIdx.x;
    const int stride = (int)blockDim.x;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) fetchNext = 0;
    queuesBegin(L, st, ctl, Q_EXTP, 0u, order, Q_EXT);
    const uint32_t n = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    uint32_t nodes = 0, prims = 0, rays = 0;
    constexpr uint32_t KINDS = SOLIDS ? KINDS_ALL : KINDS_MESH;

    bool busy = false;
    uint32_t slot = 0, local = 0;
    RayD world; world.o = splat3(0.0f); world.d = splat3(1.0f); world.tmin = 0.0f; world.tmax = 0.0f;
    RayD ray = world;                            // the ray of the level the walk is at (master space at level 2)
    f3 invD = splat3(1.0f), winvD = splat3(1.0f);
    float tmax = 0.0f;                           // the world ray's farT: the hit so far (it may GROW inside an instance set)
    float4 hit = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    int hitInst = -1;
    int cur = 0, sp = 0, level = 0;
    // level 1: BinaryBvh::trace's tMin / tMax / nearFar[2..3], the stack level the set was entered at
    float refTMin = 0.0f, refTMax = 0.0f, refFarT = 0.0f;
    int refSp = 0;
    // the leaf of the instance tree being worked off: its slots [leafNext, leafEnd) of inst_prims, the instance inside of which the walk is
    uint32_t leafNext = 0, leafEnd = 0;
    int instSp = 0, curInst = -1;
    float ltmax = 0.0f;
    float4 lhit = hit;
    bool exhausted = false;                      // wave-uniform: the queue has been handed out completely
    // COUNT: how often the wave runs each section of the turn and how many lanes have work in it (tghip_get_walk_stats, walk 0, [24 + 2 k] / [25 + 2 k]:
    // k = 0 turns / busy lanes, 1 node section, 2 of it at level 1, 3 leaf section, 4 instance leaf, 5 set entry, 6 triangle leaf, 7 pop section,
    // 8 master done, 9 instance-tree pop, 10 publish, 11 refill)
    uint32_t sec[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define INST_SECTION(k, pred) do { if (COUNT) { const unsigned long long m_ = __ballot(pred); if (m_) { sec[2*(k)]++; sec[2*(k) + 1] += (uint32_t)__popcll(m_); } } } while (0)
    for (;;) {
        unsigned long long busyMask = __ballot(busy);
        if (!exhausted && __popcll(busyMask) <= 48) {
            INST_SECTION(11, !busy);
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
                    world.o = xyz(ro); world.d = xyz(rd); world.tmin = ro.w; world.tmax = rd.w;
                    ray = world;
                    winvD = mk3(1.0f/world.d.x, 1.0f/world.d.y, 1.0f/world.d.z);
                    invD = winvD;
                    tmax = world.tmax;
                    hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
                    hitInst = -1;
                    cur = 0; sp = 0; level = 0;
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
        bool pop = false;
        INST_SECTION(0, busy);
        INST_SECTION(1, busy && cur >= 0);
        INST_SECTION(2, busy && cur >= 0 && level == 1);
        // ---- a node: the scene's and the masters' with this library's slab test, the instance tree's with the reference's ----
        if (busy && cur >= 0) {
            const float4 *nd = &at32(s.nodes, (uint32_t)cur*4u);
            const float4 n0 = ld4(nd, 0u), n1 = ld4(nd, 1u), n2 = ld4(nd, 2u), n3 = ld4(nd, 3u);
            if (COUNT) nodes++;
            const f3 lo0 = mk3(n0.x, n0.y, n0.z), hi0 = mk3(n0.w, n1.x, n1.y), lo1 = mk3(n1.z, n1.w, n2.x), hi1 = mk3(n2.y, n2.z, n2.w);
            const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            float e0, e1;
            bool h0, h1, firstIs1;
            if (level == 1) {
                h0 = refChildTest(lo0, hi0, world.o, world.d, winvD, world.tmin, refFarT, e0);
                h1 = refChildTest(lo1, hi1, world.o, world.d, winvD, world.tmin, refFarT, e1);
                firstIs1 = !(e0 < e1);                                  // `minMax[0] < minMax[1]`: the right child first on a tie
            } else {
                const float far = level == 2 ? ltmax : tmax;
                h0 = boxTest(lo0, hi0, ray, invD, far, e0);
                h1 = boxTest(lo1, hi1, ray, invD, far, e1);
                firstIs1 = e1 < e0;
            }
            float entry = 0.0f;
            if (h0 && h1) {
                stack[sp*stride] = firstIs1 ? c0 : c1;
                sp++;
                cur = firstIs1 ? c1 : c0;
                entry = firstIs1 ? e1 : e0;
            } else if (h0) { cur = c0; entry = e0; }
            else if (h1) { cur = c1; entry = e1; }
            else pop = true;
            if (level == 1 && !pop) refTMin = entry;                    // BinaryBvh::trace's tMin: the distance the ray enters the child at
        }
        {
            unsigned long long atLeaf = __ballot(busy && cur < 0 && !pop);
            unsigned long long atNode = __ballot(busy && (cur >= 0 || pop));
            // leaves are worked off when a good part of the wave waits at one, or nobody has node work left (k_trace_closest_dyn)
            if (atLeaf != 0ull && ((uint32_t)__popcll(atLeaf) >= st.leaf_batch_bvh2 || atNode == 0ull)) {
                INST_SECTION(3, busy && cur < 0 && !pop);
                INST_SECTION(4, busy && cur < 0 && !pop && level == 1);
                INST_SECTION(6, busy && cur < 0 && !pop && level != 1);
                if (busy && cur < 0 && !pop) {
                    if (level == 1) {
                        // a leaf of the instance tree: its one or two instances, one per turn
                        if (leafNext == leafEnd) { leafNext = TGHIP_LEAF_FIRST(cur); leafEnd = leafNext + TGHIP_LEAF_COUNT(cur); }
                        const uint32_t ri = s.inst_prims[leafNext];
                        leafNext++;
                        if (COUNT) prims++;
                        if (instanceReachable(s, ri, world, winvD, refTMin)) {
                            int root;
                            instanceLocalRay(s, ri, world, refTMin, PT_INF, ray, root);
                            invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
                            ltmax = PT_INF;
                            lhit = make_float4(PT_INF, 0.0f, 0.0f, __int_as_float(-1));
                            curInst = (int)ri;
                            instSp = sp;
                            stack[sp*stride] = cur;                     // (the leaf waits under the master's walk for its second instance / its end)
                            sp++;
                            cur = root;
                            level = 2;
                        } else if (leafNext == leafEnd) {
                            // the leaf's last instance cannot be hit: the leaf is done (tMax = min(tMax, ray.farT()), BinaryBvh.hpp:274-275)
                            refTMax = refMin(refTMax, tmax);
                            refFarT = refTMax;
                            leafNext = leafEnd = 0;
                            pop = true;
                        }                                               // (else: the leaf's second instance, in a turn of its own)
                    } else {
                        const uint32_t firstRec = TGHIP_LEAF_FIRST(cur), count = TGHIP_LEAF_COUNT(cur);
                        bool entered = false;
                        if (level == 0) {
                            const float4 r0 = ld4(s.recs, firstRec*3u);
                            INST_SECTION(5, TGHIP_REC_KIND(__float_as_uint(r0.w)) == TGHIP_REC_INSTANCE_SET);
                            if (TGHIP_REC_KIND(__float_as_uint(r0.w)) == TGHIP_REC_INSTANCE_SET) {   // (alone in its leaf)
                                if (COUNT) prims++;
                                const float4 r1 = ld4(s.recs, firstRec*3u + 1u), r2 = ld4(s.recs, firstRec*3u + 2u);
                                float tMin = world.tmin, tMax = tmax;
                                if (refBboxIntersection(xyz(r0), xyz(r1), world, tMin, tMax)) {
                                    refTMin = tMin; refTMax = tMax; refFarT = tmax;
                                    refSp = sp;
                                    leafNext = leafEnd = 0;
                                    cur = __float_as_int(r2.x);
                                    level = 1;
                                } else {
                                    pop = true;
                                }
                                entered = true;
                            }
                        }
                        if (!entered) {
                            for (uint32_t r = firstRec; r < firstRec + count; ++r) {
                                if (COUNT) prims++;
                                uint32_t meta;
                                if (level == 2) (void)testRecord<false, KINDS>(s, r, ray, ltmax, lhit, meta);
                                else if (testRecord<false, KINDS>(s, r, ray, tmax, hit, meta)) hitInst = -1;
                            }
                            pop = true;
                        }
                    }
                }
            }
        }
        INST_SECTION(7, busy && pop);
        INST_SECTION(8, busy && pop && level == 2 && sp == instSp + 1);
        if (busy && pop) {
            if (level == 2 && sp == instSp + 1) {
                // the master's subtree is done: a hit there REPLACES the hit so far (Instance.cpp:297-301); back to the leaf of the instance tree
                if (__float_as_int(lhit.w) >= 0) { hit = lhit; hitInst = curInst; tmax = lhit.x; }
                ray = world; invD = winvD;
                level = 1;
                sp--;
                cur = stack[sp*stride];                                 // the leaf
                if (leafNext == leafEnd) {                              // its last instance: tMax = min(tMax, ray.farT()) (BinaryBvh.hpp:274-275)
                    refTMax = refMin(refTMax, tmax);
                    refFarT = refTMax;
                    leafNext = leafEnd = 0;
                } else {
                    pop = false;                                        // the leaf's second instance, in a turn of its own
                }
            }
            INST_SECTION(9, pop && level == 1);
            if (pop && level == 1) {
                // BinaryBvh::trace's pop (:277-283): a leaf that begins behind tMax is dropped (inner nodes: refLeafEntry)
                for (;;) {
                    if (sp == refSp) { level = 0; break; }              // the set is done: on with the scene's tree (pop stays set)
                    sp--;
                    cur = stack[sp*stride];
                    if (cur >= 0) { pop = false; break; }
                    refTMin = refLeafEntry(s, TGHIP_LEAF_FIRST(cur), world.o, world.d, winvD, world.tmin);
                    if (!(refTMax < refTMin)) { pop = false; break; }
                }
            }
            INST_SECTION(10, pop && level != 1 && sp == 0);
            if (pop && level != 1) {
                if (sp == 0) {
                    // finished: publish the hit and bin the path by shading class
                    slotF4(st, A_HIT, slot) = hit;
                    slotW(st, A_EMI, slot, 3u) = __int_as_float(hitInst);
                    int ri = __float_as_int(hit.w);
                    int cls = ri < 0 ? CLS_MISS : (int)at32(s.rec_class, (uint32_t)ri);
                    queuePush(true, local, L, shadeQueue(cls));
                    busy = false;
                } else {
                    sp--;
                    cur = stack[sp*stride];
                }
            }
        }
    }
    if (COUNT && laneId() == 0) {
        unsigned long long *wk = st.stats[blockIdx.x].walk[0];
        atomicAdd(&wk[4], 1ull);
        for (int k = 0; k < 24; ++k) atomicAdd(&wk[24 + k], (unsigned long long)sec[k]);
    }
#undef INST_SECTION
    waveAddStat(&L.closest_rays, rays);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }
    queuesEnd(L, st, Q_EXT, Q_SHADE_MASK, Q_EXTP);
    if (threadIdx.x == 0) {
        ctl.closest_rays += L.closest_rays;
        if (COUNT) { st.stats[blockIdx.x].nodes_visited += L.nodes; st.stats[blockIdx.x].prims_tested += L.prims; }
    }
}

// Closest hit through the 8-wide BVH (pt_kernels.h) with dynamic ray fetch, for single-level BVH scenes.  One loop turn
// advances every busy lane by ONE memory round trip: the lane asks its walk for the next thing to look at -- a node (80
// bytes) or a primitive record (48 bytes), both inside one allocation -- loads it, and then either slab-tests the node's
// eight children or intersects the record.  Idle lanes refill from the workgroup's queue as in k_trace_closest_dyn.
// Dynamic LDS: [expanded queue, 2 B per slot][group stacks, wideDepth x 8 B per thread].
// INST: the scene has instance records -- the walk enters the masters' wide subtrees (pt_kernels.h: wideEnterInstance); the
// instance a hit was reached through goes to the spare word A_EMI.w, as k_trace_closest<.., INST> leaves it.
// (five waves per SIMD: without SLP vectorisation the closest-hit walk fits 96 VGPRs with five spilled registers outside its loop -- 811 -> 768 us per
// launch, metric's workload +0.9 %, mesh1m +1 %, profiles/r5_sweep_final_kernels.txt; PT_CLOSEST_WAVES = 4 for the A/B)
#ifndef PT_CLOSEST_WAVES
#define PT_CLOSEST_WAVES 5
#endif
#ifndef WIDE_CLOSEST_BOUNDS
#define WIDE_CLOSEST_BOUNDS __launch_bounds__(512, PT_CLOSEST_WAVES)
#endif
#ifndef WIDE_SHADOW_BOUNDS
#define WIDE_SHADOW_BOUNDS __launch_bounds__(512, 4)   /* (with whole-vector slot loads the shadow walk wants 129 VGPRs: held to the 128 of 4 waves/SIMD) */
#endif
// The end of a loop turn of the two-level (INST) walks, as an instruction of its own.  Without it every path through the turn -- fourteen
// of them in k_trace_shadow_wide<., ., INST> -- and the edge of the lanes that sit the turn out meet directly in the loop latch (one block
// with 15 predecessors and 25 phis in the optimised IR), and the code the AMDGPU backend of ROCm 7.2 generates for that latch on gfx950
// is wrong: on instances10k 44 % of the pixels lose occluders inside instances, differently from run to run (which lanes share a wave
// depends on the order the waves fetch in), while the BVH2 walk, the variant that counts its visits and every variant with ANY
// side-effecting instruction at this place (a counter, s_nop, an empty asm) agree bit for bit; waits and fences anywhere else change
// nothing, so it is not a memory-ordering problem (tools/repro_latch_miscompile.py, profiles/r3_latch_miscompile.txt).
// With the join the latch has two predecessors.  Costs nothing: no instruction is emitted.
#define PT_TURN_JOIN() asm volatile("" ::: "memory")
// The fetch of a decoupled turn: the lane's record (three rows) and its node (five rows), eight independent loads issued back to back.
// PT_WALK_FETCH_SAFE = 1 (round 5): every lane loads, from entry 0 where it has nothing to fetch -- no divergent region around the loads.
// With the loads inside `if (hasRec)` / `if (hasNode)` the closest-hit kernel waited for the record's first row (s_waitcnt vmcnt(2) + three
// copies out of the region) BEFORE it issued the node's loads: two memory round trips per turn where one was meant.  The through-float4 record
// reads stay narrowable (pt_math.h: ld4).  PT_LDS_TOP = 1 compiles the top-of-tree-in-LDS experiment of round 3 back in (lds_nodes option).
#ifndef PT_WALK_FETCH_SAFE
#define PT_WALK_FETCH_SAFE 1
#endif
#ifndef PT_LDS_TOP
#define PT_LDS_TOP 0
#endif
#if PT_WALK_FETCH_SAFE
#define PT_WALK_FETCH(s, st, r0, r1, r2, nd, hasRec, recIdx, hasNode, nodeIdx, wr, topCount, ldsTop) do { \
        const uint32_t recAt_ = (hasRec) ? (recIdx)*3u : 0u; \
        const uint32_t nodeAt_ = (hasNode) ? wideNodeOff(s, nodeIdx) : 0u; \
        r0 = at32((s).recs, recAt_); r1 = at32((s).recs, recAt_ + 1u); r2 = at32((s).recs, recAt_ + 2u); \
        if (PT_LDS_TOP && (hasNode) && (nodeIdx) < (topCount)) wideNodeFetch(nd, ldsTop, nodeAt_, wr); \
        else wideNodeFetch(nd, reinterpret_cast<const char *>((s).wide), nodeAt_, wr); \
    } while (0)
#else
#define PT_WALK_FETCH(s, st, r0, r1, r2, nd, hasRec, recIdx, hasNode, nodeIdx, wr, topCount, ldsTop) do { \
        if (hasRec) { r0 = at32((s).recs, (recIdx)*3u + 0u); r1 = at32((s).recs, (recIdx)*3u + 1u); r2 = at32((s).recs, (recIdx)*3u + 2u); } \
        if (hasNode) { \
            if ((nodeIdx) < (topCount)) wideNodeFetch(nd, ldsTop, wideNodeOff(s, nodeIdx), wr); \
            else                        wideNodeFetch(nd, reinterpret_cast<const char *>((s).wide), wideNodeOff(s, nodeIdx), wr); \
        } \
    } while (0)
#endif
#ifndef PT_LATE_PUBLISH
#define PT_LATE_PUBLISH 1     /* the decoupled walks: after the queue ran dry, finished walks publish more than eight at a time (0 for the A/B) */
#endif
#ifndef PT_SHADOW_PREP
#define PT_SHADOW_PREP 1      /* k_trace_shadow_fast: a slot's second ray prepared in the refill block (traceShadowFastBody); 0 for the A/B */
#endif
// busy lanes at or below which a wave of the decoupled walks takes new rays from its workgroup's queue (swept in round 5, r5_sweep_final_kernels.txt: 24 / 32 / 40 / 48 / 56 -- 40 is the shadow walk's optimum, the closest-hit walk is level between 40 and 48)
#ifndef PT_REFILL_AT
#define PT_REFILL_AT 40
#endif
// DECOUPLED (single-level scenes): a turn tests the lane's next pending record AND visits its next node -- the walk does not wait for
// the records of the node visited last before it moves on (their outcome only tightens tmax, never what is visited next), so a ray
// needs about max(nodes, records) turns instead of their sum; a node visited before an earlier node's records have shortened the ray
// may report a few children more (conservative: hits unchanged, visit counts a little above the sequential walk's).
// (the kernel's body as a function of the workgroup's LDS objects: k_trace_closest_wide below and the tail kernel, k_tail, run it)
template<bool COUNT, bool SOLIDS, bool INST, bool DECOUPLED, int NTS = PT_NT_TRAV>
PT_DEV void traceClosestWideBody(const DeviceScene &s, const PathState &st, BlockLds &L, uint32_t &fetchNext, int *ldsDyn)
{
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    uint2 *stack = reinterpret_cast<uint2 *>(ldsDyn + (st.slots_per_block >> 1)) + threadIdx.x;
    const int stride = (int)blockDim.x;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) fetchNext = 0;
    const unsigned long long wpStart = COUNT ? wall_clock64() : 0ull;
    // DECOUPLED: the top of the tree in LDS, behind the stacks (PathState::lds_nodes; queuesBegin's barriers publish the copy)
    const uint32_t topCount = DECOUPLED ? st.lds_nodes : 0u;
    const char *ldsTop = reinterpret_cast<const char *>(ldsDyn) + st.slots_per_block*2u + st.wide_depth*blockDim.x*8u;
    if constexpr (DECOUPLED) {
        float4 *dst = reinterpret_cast<float4 *>(const_cast<char *>(ldsTop));
        for (uint32_t i = threadIdx.x; i < topCount*s.wide_stride/16u; i += blockDim.x)
            dst[i] = s.wide[i];
    }
    queuesBegin(L, st, ctl, Q_EXTP, 0u, order, Q_EXT);
    const uint32_t n = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    uint32_t nodes = 0, prims = 0, rays = 0;
    uint32_t turns = 0, dryTurns = 0;           // COUNT: loop turns of this wave, and those after the queue ran dry (lane utilisation, tools)

    bool busy = false;
    uint32_t slot = 0, local = 0;
    RayD ray; ray.o = splat3(0.0f); ray.d = splat3(1.0f); ray.tmin = 0.0f; ray.tmax = 0.0f;
    WideRay wr; wr.idir = splat3(1.0f); wr.octInv = 0u;
    WideState w;
    wideStart(w);
    float tmax = 0.0f;
    float4 hit = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    int hitInst = -1;
    bool pendingPublish = false;                 // DECOUPLED: the walk is over, its hit is published at the next refill (below)
    bool exhausted = false;                      // wave-uniform: the queue has been handed out completely
    uint32_t age = 0;                            // turns this lane's walk has had in this launch (PathState::suspend_turns)
    const bool maySuspend = !INST && st.suspend_lanes != 0u && n >= st.suspend_min_queue;
    WALK_PROF_DECL;
    for (;;) {
        unsigned long long busyMask = __ballot(busy);
        if constexpr (DECOUPLED) {
            // Finished walks publish their hit -- and bin their path by the shading class of the record hit, a dependent load -- together,
            // right before the lanes are refilled (once the queue is dry: in the turn they finish): one memory round trip per refill
            // instead of one in nearly every turn, sat out by the whole wave.
            // (round 6: once the queue is dry the finished walks wait until more than eight of them can publish together, or nothing else is
            // left -- eight or fewer enabled lanes issue VALU instructions at a quarter of the rate, profiles/r6_ubench_lane_masks.txt)
            const unsigned long long pendMask = __ballot(pendingPublish);
            if (pendMask != 0ull && ((!exhausted && __popcll(busyMask) <= PT_REFILL_AT) || (exhausted && (!PT_LATE_PUBLISH || __popcll(pendMask) > 8 || busyMask == 0ull)))) {
                WALK_SECTION(wpPubs, wpPubLanes, pendingPublish);
                if (pendingPublish) {
                    slotF4<NTS>(st, A_HIT, slot) = hit;
                    const int ri = __float_as_int(hit.w);
                    const int cls = ri < 0 ? CLS_MISS : (int)at32(s.rec_class, (uint32_t)ri);
                    queuePush(true, local, L, shadeQueue(cls));
                    pendingPublish = false;
                }
            }
        }
        if (!exhausted && __popcll(busyMask) <= PT_REFILL_AT) {
            unsigned long long want = ~busyMask;
            uint32_t lane = laneId();
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&fetchNext, (uint32_t)__popcll(want));
            base = __shfl(base, 0);
            WALK_SECTION(wpRefills, wpRefillLanes, !busy && base + __popcll(want & ((1ull << lane) - 1ull)) < n);
            if (!busy) {
                uint32_t i = base + __popcll(want & ((1ull << lane) - 1ull));
                if (i < n) {
                    local = order[i];
                    slot = first + local;
                    float4 ro = slotF4<NTS>(st, A_RAY_O, slot), rd = slotF4<NTS>(st, A_RAY_D, slot);
                    ray.o = xyz(ro); ray.d = xyz(rd); ray.tmin = ro.w; ray.tmax = rd.w;
                    bool resumed = false;
                    if constexpr (!INST) {
                        resumed = (__float_as_uint(ro.w) & WALK_SUSPENDED_BIT) != 0u;   // a walk an earlier launch suspended (below)
                        ray.tmin = __uint_as_float(__float_as_uint(ro.w) & ~WALK_SUSPENDED_BIT);
                    }
                    wr = wideRaySetup(ray);
                    if (!resumed) {
                        wideStart(w);
                        tmax = ray.tmax;
                        hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
                        if constexpr (DECOUPLED && !INST) {
                            if (s.hoisted_rec >= 0) {            // the scene's one quad, before the walk (DeviceScene::hoisted_rec; uniform loads)
                                uint32_t meta;
                                if (COUNT) prims++;
                                (void)testRecord<true, KIND_BIT(TGHIP_REC_QUAD)>(s, (uint32_t)s.hoisted_rec, ray, tmax, hit, meta);
                            }
                        }
                        rays++;
                        if (COUNT) wpRays++;
                    } else {
                        walkRestore(st, slot, w, stack, stride);
                        if (COUNT) wpResumed++;
                        hit = slotF4<NTS>(st, A_HIT, slot);                      // the best hit so far
                        tmax = hit.x;
                        slotW(st, A_RAY_O, slot, 3u) = ray.tmin;             // (the ray is an ordinary one again)
                    }
                    hitInst = -1;
                    age = 0;
                    busy = true;
                }
            }
            if (base + (uint32_t)__popcll(want) >= n) {
                exhausted = true;
                if (COUNT) wpDry = wall_clock64();
            }
            busyMask = __ballot(busy);
        }
        if constexpr (!INST) {
            // The queue is dry and this wave is down to its last, longest walks: those that have had their turns in this launch are
            // suspended -- state to the slot's walk arrays, the ray back on the extension queue -- and continue in the next launch
            // in a full wave (PathState::suspend_*).
            if (maySuspend && exhausted && (uint32_t)__popcll(busyMask) <= st.suspend_lanes) {
                if (busy && age >= st.suspend_turns) {
                    walkSave(st, slot, w, stack, stride);
                    if (COUNT) wpSuspended++;
                    slotF4<NTS>(st, A_HIT, slot) = hit;
                    slotW(st, A_RAY_O, slot, 3u) = __uint_as_float(__float_as_uint(ray.tmin) | WALK_SUSPENDED_BIT);
                    queuePush(true, local, L, Q_EXT);
                    busy = false;
                }
                busyMask = __ballot(busy);
            }
        }
        if (busyMask == 0ull) {
            if (DECOUPLED && __ballot(pendingPublish) != 0ull)
                continue;                        // (the walks that ended in the last turn: published at the top of the loop)
            break;
        }
        age++;
        if (COUNT) { turns++; dryTurns += exhausted ? 1u : 0u; if (exhausted) wpBusyDry += (uint32_t)__popcll(busyMask); else wpBusy += (uint32_t)__popcll(busyMask); }
        if constexpr (!INST && DECOUPLED) {
            uint32_t recIdx = 0, nodeIdx = 0;
            bool hasRec = false, hasNode = false, finished = false;
            if (busy) {
                if (w.triMask == 0u && w.tri2Mask != 0u) { w.triBase = w.tri2Base; w.triMask = w.tri2Mask; w.triValid = w.tri2Valid; w.tri2Mask = 0u; }
                if (w.triMask) {
                    const uint32_t b = (uint32_t)__ffs((int)w.triMask) - 1u;
                    recIdx = w.triBase + (uint32_t)__popc(w.triValid & ((1u << b) - 1u));
                    w.triMask &= w.triMask - 1u;
                    hasRec = true;
                }
                if (w.tri2Mask == 0u)            // (room for the records of the node visited now)
                    hasNode = wideNextNode(w, wr.octInv, stack, stride, nodeIdx);
            }
            float4 r0, r1, r2;
            WideNodeRegs nd;
            PT_WALK_FETCH(s, st, r0, r1, r2, nd, hasRec, recIdx, hasNode, nodeIdx, wr, topCount, ldsTop);
            WALK_SECTION(wpRecTurns, wpRecLanes, hasRec);
            WALK_SECTION(wpNodeTurns, wpNodeLanes, hasNode);
            if (hasRec) {
                if (COUNT) prims++;
                uint32_t meta;
                const bool accepted = testRecordLoaded<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, recIdx, r0, r1, r2, ray, tmax, hit, meta);
                if (COUNT && accepted) wpHits++;
            }
            if (hasNode) {
                if (COUNT) nodes++;
                const uint32_t ob = w.triBase, om = w.triMask, ov = w.triValid;
                wideVisit(w, nd, ray.o, wr, ray.tmin, tmax, s.hoisted_rec >= 0);
                if (om) { w.tri2Base = w.triBase; w.tri2Mask = w.triMask; w.tri2Valid = w.triValid; w.triBase = ob; w.triMask = om; w.triValid = ov; }
            }
            if (busy && wideWalkOver(w)) {       // (in the turn that looked at the walk's last record / node)
                busy = false;
                pendingPublish = true;
            }
            (void)finished;
        } else if constexpr (!INST) {
            // What the lane looks at this turn: its next record, its next node, or -- when that record is the last one of the node
            // visited before -- BOTH: the node to visit next does not depend on the record's outcome (the group's order is fixed),
            // only its slab tests do, and they run after the record test; so the walk is the one wideNext defines, a record-bearing
            // node just costs one turn less.  (leaf_batch = 9 switches the pairing off.)
            uint32_t recIdx = 0, nodeIdx = 0;
            bool hasRec = false, hasNode = false, finished = false;
            if (busy) {
                uint32_t idx = 0;
                const int what = wideNext<false>(w, wr.octInv, stack, stride, idx);
                if (what == 1) {
                    hasRec = true; recIdx = idx;
                    if (w.triMask == 0u && st.leaf_batch != 9u) {
                        const int next = wideNext<false>(w, wr.octInv, stack, stride, idx);   // a node, or the end of the walk
                        if (next == 2) { hasNode = true; nodeIdx = idx; } else finished = true;
                    }
                } else if (what == 2) { hasNode = true; nodeIdx = idx; }
                else finished = true;
            }
            float4 r0, r1, r2;
            WideNodeRegs nd;
            if (hasRec) { r0 = at32(s.recs, recIdx*3u + 0u); r1 = at32(s.recs, recIdx*3u + 1u); r2 = at32(s.recs, recIdx*3u + 2u); }
            if (hasNode) wideNodeFetch(nd, reinterpret_cast<const char *>(s.wide), wideNodeOff(s, nodeIdx), wr);
            if (hasRec) {
                if (COUNT) prims++;
                uint32_t meta;
                (void)testRecordLoaded<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, recIdx, r0, r1, r2, ray, tmax, hit, meta);
            }
            if (hasNode) {
                if (COUNT) nodes++;
                wideVisit(w, nd, ray.o, wr, ray.tmin, tmax);
            }
            if (finished) {
                // publish the hit and bin the path by shading class
                slotF4<NTS>(st, A_HIT, slot) = hit;
                int ri = __float_as_int(hit.w);
                int cls = ri < 0 ? CLS_MISS : (int)at32(s.rec_class, (uint32_t)ri);
                queuePush(true, local, L, shadeQueue(cls));
                busy = false;
            }
        } else {
            uint32_t idx = 0;
            int what = 0;
            if (busy) {
                // Phase vote (st.leaf_batch != 1): the slab tests of a node and the intersection of a record are ~220 and ~110
                // VALU instructions that every lane of the wave sits through, so a turn runs only the kind the majority of
                // the busy lanes wants next; the others keep their request for a later turn.
                const bool wantsRecord = w.triMask != 0u;
                bool go = true;
                if (st.leaf_batch != 1u) {
                    const uint32_t nRec = (uint32_t)__popcll(__ballot(wantsRecord)), nNode = (uint32_t)__popcll(__ballot(!wantsRecord));
                    go = (nRec*st.leaf_batch >= nNode*2u) == wantsRecord;
                }
                if (go) {
                    what = wideNext<INST>(w, wr.octInv, stack, stride, idx);
                    if (what == 0) {
                        // finished: publish the hit and bin the path by shading class
                        slotF4<NTS>(st, A_HIT, slot) = hit;
                        if (INST) slotW(st, A_EMI, slot, 3u) = __int_as_float(hitInst);
                        int ri = __float_as_int(hit.w);
                        int cls = ri < 0 ? CLS_MISS : (int)at32(s.rec_class, (uint32_t)ri);
                        queuePush(true, local, L, shadeQueue(cls));
                        busy = false;
                    }
                }
            }
            if (INST && what == 3) {
                // the master's subtree is done: back to the world-space ray (distances along it did not change)
                float4 ro = slotF4<NTS>(st, A_RAY_O, slot), rd = slotF4<NTS>(st, A_RAY_D, slot);
                ray.o = xyz(ro); ray.d = xyz(rd);
                wr = wideRaySetup(ray);
                w.curInst = -1;
            } else if (what != 0) {
                // one address per lane: a node or a record, both behind s.wide
                const uint32_t off = what != 1 ? wideNodeOff(s, idx) : s.recs_offset + idx*48u;
                const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.wide) + (size_t)off);
                float4 q0 = p[0], q1 = p[1], q2 = p[2];
                if (what == 2) {
                    WideNodeRegs nd;
                    wideNodeFetchRest(nd, reinterpret_cast<const char *>(s.wide), off, wr, q0, q1, q2);
                    if (COUNT) nodes++;
                    wideVisit(w, nd, ray.o, wr, ray.tmin, tmax);
                } else if (INST && what == 4) {
                    wideResumeRecords(w, idx, q1);
                } else {
                    if (COUNT) prims++;
                    if (INST && TGHIP_REC_KIND(__float_as_uint(q0.w)) == TGHIP_REC_INSTANCE) {
                        wideEnterInstance(w, stack, stride, idx, q0, q1, q2, ray, wr);
                    } else {
                        uint32_t meta;
                        if (testRecordLoaded<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, idx, q0, q1, q2, ray, tmax, hit, meta))
                            hitInst = w.curInst;
                    }
                }
            }
            PT_TURN_JOIN();
            }
    }
    if (COUNT) wpLoopEnd = wall_clock64();
    waveAddStat(&L.closest_rays, rays);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }
    queuesEnd(L, st, Q_EXT, Q_SHADE_MASK, Q_EXTP);
    if (threadIdx.x == 0) {
        ctl.closest_rays += L.closest_rays;
        if (COUNT) { st.stats[blockIdx.x].nodes_visited += L.nodes; st.stats[blockIdx.x].prims_tested += L.prims; }
    }
    if (COUNT && laneId() == 0) { atomicAdd(&st.stats[blockIdx.x].prof[10], (unsigned long long)turns); atomicAdd(&st.stats[blockIdx.x].prof[11], (unsigned long long)dryTurns); }
    if (COUNT) WALK_PROF_FLUSH(0, turns - dryTurns, dryTurns);
}
template<bool COUNT, bool SOLIDS = true, bool INST = false, bool DECOUPLED = false>
__global__ WIDE_CLOSEST_BOUNDS void k_trace_closest_wide(DeviceScene s, PathState st)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    traceClosestWideBody<COUNT, SOLIDS, INST, DECOUPLED>(s, st, L, fetchNext, ldsDyn);
}
// The same launch with k_finish's work of the PREVIOUS iteration in front: every workgroup first finalises the paths that ended at the last
// vertex and regenerates their slots (finishBody, below), then traces its extension rays, the fresh camera rays among them -- one launch per
// part and iteration less, and the streaming of the regeneration runs inside the issue-bound walk's launch.  The shim launches the stand-alone
// k_finish only before a host check (the liveness report) and before k_tail.  Single-level scenes on the decoupled walk.
// EXT = false: the pass carries none of nextPath's run-time extras (Sobol' sampler, SampleRecords, auxiliary outputs, per-sample output, thin lens, media:
// pp.flags == 0 -- the metric's passes): the finish in front of the walk is nextPath's lean variant, as in the specialised shading kernels.
template<int NTS = PT_NT_TRAV, bool EXT = true>
PT_DEV bool finishBody(const DeviceScene &s, const PathState &st, const PassParams &pp, BlockLds &L, unsigned short *order);
template<bool COUNT, bool SOLIDS, bool EXT = true>
__global__ WIDE_CLOSEST_BOUNDS void k_finish_trace_closest_wide(DeviceScene s, PathState st, PassParams pp)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    (void)finishBody<PT_NT_TRAV, EXT>(s, st, pp, L, reinterpret_cast<unsigned short *>(ldsDyn));   // (the queue area of the dynamic LDS: slots_per_block entries)
    __syncthreads();
    traceClosestWideBody<COUNT, SOLIDS, false, true>(s, st, L, fetchNext, ldsDyn);
}

// Round 6: k_trace_closest_inst with the masters' subtrees walked through the 8-wide BVH (the DECOUPLED walk of k_trace_closest_wide: a pending
// record and the next node per turn), and the turn split into two phases a wave votes on.  What the counters said about the kernel above
// (profiles/r6_sq_counters_instances10k.json, r6_bench_instances10k_sections.json): 0.22 of the lanes of an issued VALU instruction enabled -- every turn
// runs the node step of two box tests' flavours, the leaf code of three kinds and the pop code, each for the handful of lanes that are there --, and
// 32.8 BVH2 nodes per ray, nearly all of them inside masters.  Inside a master the walk is a plain nearest-hit query (Instance.cpp:296-303 constrains
// the order BETWEEN instances only: farT = infinity going in, the master's nearest hit REPLACES the hit so far), so it may be any correct one:
//   phase M  lanes inside a master (level 2): one decoupled turn of the wide walk over the master's wide subtree (instance record c[2]) with a stack
//            of its own (uint2 entries: behind the BVH2 stack in the dynamic LDS); a master that is done hands its hit over right there;
//   phase T  lanes in the scene's BVH2 (level 0) and in the reference's tree over the instances (level 1): the statements of the kernel above --
//            the reference's child test and pop rule, leaf by leaf in the reference's order, one instance per turn.
// A phase runs when it has more lanes than the other or at least PathState::inst_phase_min of them; the others keep their state for a later turn.
// Same hits as the kernel above wherever a master's nearest hit is unique (ties inside a master follow the wide walk's order instead of the
// BVH2's; the reference's own order there is Embree's, which neither restates: DESIGN.md "ties").
// (walk statistics, tghip_get_walk_stats walk 0: [22] = wave launches of THIS kernel, [23] = wide nodes visited, [24 + 2 k] / [25 + 2 k] = runs / lanes of
//  k = 0 turns / busy lanes, 1 phase M, 2 its record test, 3 its node visit, 4 master done, 5 phase T, 6 node step, 7 leaf section, 8 instance leaf,
//  9 master entered, 10 pop section, 11 refill)
template<bool COUNT, bool SOLIDS = true>
__global__ __launch_bounds__(512) void k_trace_closest_instw(DeviceScene s, PathState st)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLdsSmall L;
    __shared__ uint32_t fetchNext;
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    int *stack = ldsDyn + (st.slots_per_block >> 1) + threadIdx.x;
    const int stride = (int)blockDim.x;
    // the masters' group stack behind the BVH2 stack (8-byte aligned: inst_tree_depth*blockDim is even, the queue area rounded up)
    uint2 *wstack = reinterpret_cast<uint2 *>(ldsDyn + (((st.slots_per_block >> 1) + st.inst_tree_depth*blockDim.x + 1u) & ~1u)) + threadIdx.x;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) fetchNext = 0;
    queuesBegin(L, st, ctl, Q_EXTP, 0u, order, Q_EXT);
    const uint32_t n = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    uint32_t nodes = 0, wnodes = 0, prims = 0, rays = 0;
    constexpr uint32_t KINDS = SOLIDS ? KINDS_ALL : KINDS_MESH;
    const uint32_t phaseMin = st.inst_phase_min;

    bool busy = false;
    uint32_t slot = 0, local = 0;
    RayD world; world.o = splat3(0.0f); world.d = splat3(1.0f); world.tmin = 0.0f; world.tmax = 0.0f;
    f3 winvD = splat3(1.0f);
    RayD ray = world;                            // level 2: the ray in the master's space
    WideRay wr; wr.idir = splat3(1.0f); wr.octInv = 0u;
    WideState w;
    wideStart(w);
    float tmax = 0.0f;                           // the world ray's farT: the hit so far (it may GROW inside an instance set)
    float4 hit = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    int hitInst = -1;
    int cur = 0, sp = 0, level = 0;
    float refTMin = 0.0f, refTMax = 0.0f, refFarT = 0.0f;   // level 1: BinaryBvh::trace's tMin / tMax / nearFar[2..3]
    int refSp = 0;                               // ... and the stack level the set was entered at
    uint32_t leafNext = 0, leafEnd = 0;          // the leaf of the instance tree being worked off: its slots [leafNext, leafEnd) of inst_prims
    int curInst = -1;                            // the instance inside of which the walk is
    float ltmax = 0.0f;
    float4 lhit = hit;
    bool popPending = false;                     // a lane of phase T that has to pop first (its leaf's last master just ended in phase M)
    bool exhausted = false;                      // wave-uniform: the queue has been handed out completely
    uint32_t sec[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define INST_SECTION(k, pred) do { if (COUNT) { const unsigned long long m_ = __ballot(pred); if (m_) { sec[2*(k)]++; sec[2*(k) + 1] += (uint32_t)__popcll(m_); } } } while (0)
    for (;;) {
        unsigned long long busyMask = __ballot(busy);
        if (!exhausted && (uint32_t)__popcll(busyMask) <= st.inst_refill_at) {
            INST_SECTION(11, !busy);
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
                    world.o = xyz(ro); world.d = xyz(rd); world.tmin = ro.w; world.tmax = rd.w;
                    winvD = mk3(1.0f/world.d.x, 1.0f/world.d.y, 1.0f/world.d.z);
                    tmax = world.tmax;
                    hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
                    hitInst = -1;
                    cur = 0; sp = 0; level = 0;
                    popPending = false;
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
        // ---- the vote ----
        const bool inMaster = busy && level == 2;
        const uint32_t nM = (uint32_t)__popcll(__ballot(inMaster)), nT = (uint32_t)__popcll(busyMask) - nM;
        const bool runM = nM != 0u && (nM >= nT || nM >= phaseMin);
        const bool runT = nT != 0u && (nT > nM || nT >= phaseMin);
        INST_SECTION(0, busy);
        // ---- phase M: a decoupled turn of the master's wide walk ----
        if (runM) {
            INST_SECTION(1, inMaster);
            uint32_t recIdx = 0, nodeIdx = 0;
            bool hasRec = false, hasNode = false;
            if (inMaster) {
                if (w.triMask == 0u && w.tri2Mask != 0u) { w.triBase = w.tri2Base; w.triMask = w.tri2Mask; w.triValid = w.tri2Valid; w.tri2Mask = 0u; }
                if (w.triMask) {
                    const uint32_t b = (uint32_t)__ffs((int)w.triMask) - 1u;
                    recIdx = w.triBase + (uint32_t)__popc(w.triValid & ((1u << b) - 1u));
                    w.triMask &= w.triMask - 1u;
                    hasRec = true;
                }
                if (w.tri2Mask == 0u)            // (room for the records of the node visited now)
                    hasNode = wideNextNode(w, wr.octInv, wstack, stride, nodeIdx);
            }
            float4 r0, r1, r2;
            WideNodeRegs nd;
            PT_WALK_FETCH(s, st, r0, r1, r2, nd, hasRec, recIdx, hasNode, nodeIdx, wr, 0u, reinterpret_cast<const char *>(s.wide));
            INST_SECTION(2, hasRec);
            INST_SECTION(3, hasNode);
            if (hasRec) {
                if (COUNT) prims++;
                uint32_t meta;
                (void)testRecordLoaded<false, KINDS>(s, recIdx, r0, r1, r2, ray, ltmax, lhit, meta);
            }
            if (hasNode) {
                if (COUNT) wnodes++;
                const uint32_t ob = w.triBase, om = w.triMask, ov = w.triValid;
                wideVisit(w, nd, ray.o, wr, ray.tmin, ltmax);
                if (om) { w.tri2Base = w.triBase; w.tri2Mask = w.triMask; w.tri2Valid = w.triValid; w.triBase = ob; w.triMask = om; w.triValid = ov; }
            }
            const bool masterDone = inMaster && wideWalkOver(w);
            INST_SECTION(4, masterDone);
            if (masterDone) {
                // the master's subtree is done: a hit there REPLACES the hit so far (Instance.cpp:297-301); back to the leaf of the instance tree (`cur`)
                if (__float_as_int(lhit.w) >= 0) { hit = lhit; hitInst = curInst; tmax = lhit.x; }
                level = 1;
                if (leafNext == leafEnd) {                              // its last instance: tMax = min(tMax, ray.farT()) (BinaryBvh.hpp:274-275)
                    refTMax = refMin(refTMax, tmax);
                    refFarT = refTMax;
                    leafNext = leafEnd = 0;
                    popPending = true;
                }                                                       // (else: the leaf's second instance, in a turn of phase T)
            }
        }
        // ---- phase T: the scene's BVH2 and the reference's tree over the instances (k_trace_closest_inst's statements) ----
        if (runT) {
            const bool inTree = busy && level != 2;
            INST_SECTION(5, inTree);
            bool pop = inTree && popPending;
            popPending = inTree ? false : popPending;
            INST_SECTION(6, inTree && !pop && cur >= 0);
            if (inTree && !pop && cur >= 0) {
                const float4 *nd = &at32(s.nodes, (uint32_t)cur*4u);
                const float4 n0 = ld4(nd, 0u), n1 = ld4(nd, 1u), n2 = ld4(nd, 2u), n3 = ld4(nd, 3u);
                if (COUNT) nodes++;
                const f3 lo0 = mk3(n0.x, n0.y, n0.z), hi0 = mk3(n0.w, n1.x, n1.y), lo1 = mk3(n1.z, n1.w, n2.x), hi1 = mk3(n2.y, n2.z, n2.w);
                const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                float e0, e1;
                bool h0, h1, firstIs1;
                if (level == 1) {
                    h0 = refChildTest(lo0, hi0, world.o, world.d, winvD, world.tmin, refFarT, e0);
                    h1 = refChildTest(lo1, hi1, world.o, world.d, winvD, world.tmin, refFarT, e1);
                    firstIs1 = !(e0 < e1);                              // `minMax[0] < minMax[1]`: the right child first on a tie
                } else {
                    h0 = boxTest(lo0, hi0, world, winvD, tmax, e0);
                    h1 = boxTest(lo1, hi1, world, winvD, tmax, e1);
                    firstIs1 = e1 < e0;
                }
                float entry = 0.0f;
                if (h0 && h1) {
                    stack[sp*stride] = firstIs1 ? c0 : c1;
                    sp++;
                    cur = firstIs1 ? c1 : c0;
                    entry = firstIs1 ? e1 : e0;
                } else if (h0) { cur = c0; entry = e0; }
                else if (h1) { cur = c1; entry = e1; }
                else pop = true;
                if (level == 1 && !pop) refTMin = entry;                // BinaryBvh::trace's tMin: the distance the ray enters the child at
            }
            {
                const unsigned long long atLeaf = __ballot(inTree && cur < 0 && !pop);
                const unsigned long long atNode = __ballot(inTree && (cur >= 0 || pop));
                // leaves are worked off when a good part of the wave waits at one, or nobody has node work left (k_trace_closest_dyn)
                if (atLeaf != 0ull && ((uint32_t)__popcll(atLeaf) >= st.leaf_batch_bvh2 || atNode == 0ull)) {
                    INST_SECTION(7, inTree && cur < 0 && !pop);
                    INST_SECTION(8, inTree && cur < 0 && !pop && level == 1);
                    if (inTree && cur < 0 && !pop) {
                        if (level == 1) {
                            // a leaf of the instance tree: its one or two instances, one per turn
                            if (leafNext == leafEnd) { leafNext = TGHIP_LEAF_FIRST(cur); leafEnd = leafNext + TGHIP_LEAF_COUNT(cur); }
                            const uint32_t ri = s.inst_prims[leafNext];
                            leafNext++;
                            if (COUNT) prims++;
                            const bool reach = instanceReachable(s, ri, world, winvD, refTMin);
                            INST_SECTION(9, reach);
                            if (reach) {
                                // the ray in the master's space (Instance.cpp:295-296), nearT = the leaf's entry distance, farT = infinity
                                const float4 q0 = ld4(s.recs, ri*3u + 0u), q1 = ld4(s.recs, ri*3u + 1u), q2 = ld4(s.recs, ri*3u + 2u);
                                const f3 qc = -xyz(q1);                     // conjugate(): the inverse rotation
                                ray.o = quatRotate(q1.w, qc, world.o - xyz(q0));
                                ray.d = quatRotate(q1.w, qc, world.d);
                                ray.tmin = refTMin; ray.tmax = PT_INF;
                                wr = wideRaySetup(ray);
                                wideStart(w);
                                w.node = (int)__float_as_uint(q2.z);        // the root of the master's wide subtree
                                ltmax = PT_INF;
                                lhit = make_float4(PT_INF, 0.0f, 0.0f, __int_as_float(-1));
                                curInst = (int)ri;
                                level = 2;                                  // (`cur` stays the leaf: the master's walk has a stack of its own)
                            } else if (leafNext == leafEnd) {
                                // the leaf's last instance cannot be hit: the leaf is done (tMax = min(tMax, ray.farT()), BinaryBvh.hpp:274-275)
                                refTMax = refMin(refTMax, tmax);
                                refFarT = refTMax;
                                leafNext = leafEnd = 0;
                                pop = true;
                            }                                               // (else: the leaf's second instance, in a turn of its own)
                        } else {
                            const uint32_t firstRec = TGHIP_LEAF_FIRST(cur), count = TGHIP_LEAF_COUNT(cur);
                            bool entered = false;
                            const float4 r0 = ld4(s.recs, firstRec*3u);
                            if (TGHIP_REC_KIND(__float_as_uint(r0.w)) == TGHIP_REC_INSTANCE_SET) {   // (alone in its leaf)
                                if (COUNT) prims++;
                                const float4 r1 = ld4(s.recs, firstRec*3u + 1u), r2 = ld4(s.recs, firstRec*3u + 2u);
                                float tMin = world.tmin, tMax = tmax;
                                if (refBboxIntersection(xyz(r0), xyz(r1), world, tMin, tMax)) {
                                    refTMin = tMin; refTMax = tMax; refFarT = tmax;
                                    refSp = sp;
                                    leafNext = leafEnd = 0;
                                    cur = __float_as_int(r2.x);
                                    level = 1;
                                } else {
                                    pop = true;
                                }
                                entered = true;
                            }
                            if (!entered) {
                                for (uint32_t r = firstRec; r < firstRec + count; ++r) {
                                    if (COUNT) prims++;
                                    uint32_t meta;
                                    if (testRecord<false, KINDS>(s, r, world, tmax, hit, meta)) hitInst = -1;
                                }
                                pop = true;
                            }
                        }
                    }
                }
            }
            INST_SECTION(10, inTree && pop);
            if (inTree && level != 2 && pop) {
                if (level == 1) {
                    // BinaryBvh::trace's pop (:277-283): a leaf that begins behind tMax is dropped (inner nodes: refLeafEntry)
                    for (;;) {
                        if (sp == refSp) { level = 0; break; }          // the set is done: on with the scene's tree (pop stays set)
                        sp--;
                        cur = stack[sp*stride];
                        if (cur >= 0) { pop = false; break; }
                        refTMin = refLeafEntry(s, TGHIP_LEAF_FIRST(cur), world.o, world.d, winvD, world.tmin);
                        if (!(refTMax < refTMin)) { pop = false; break; }
                    }
                }
                if (pop) {                                              // (level 0)
                    if (sp == 0) {
                        // finished: publish the hit and bin the path by shading class
                        slotF4(st, A_HIT, slot) = hit;
                        slotW(st, A_EMI, slot, 3u) = __int_as_float(hitInst);
                        int ri = __float_as_int(hit.w);
                        int cls = ri < 0 ? CLS_MISS : (int)at32(s.rec_class, (uint32_t)ri);
                        queuePush(true, local, L, shadeQueue(cls));
                        busy = false;
                    } else {
                        sp--;
                        cur = stack[sp*stride];
                    }
                }
            }
        }
        PT_TURN_JOIN();
    }
    if (COUNT && laneId() == 0) {
        unsigned long long *wk = st.stats[blockIdx.x].walk[0];
        atomicAdd(&wk[4], 1ull);
        atomicAdd(&wk[22], 1ull);
        for (int k = 0; k < 24; ++k) atomicAdd(&wk[24 + k], (unsigned long long)sec[k]);
    }
#undef INST_SECTION
    waveAddStat(&L.closest_rays, rays);
    if (COUNT) {
        waveAddStat(&L.nodes, nodes + wnodes); waveAddStat(&L.prims, prims);
        uint32_t wn = wnodes;
        for (int off = 32; off > 0; off >>= 1) wn += __shfl_down(wn, off);
        if (laneId() == 0) atomicAdd(&st.stats[blockIdx.x].walk[0][23], (unsigned long long)wn);
    }
    queuesEnd(L, st, Q_EXT, Q_SHADE_MASK, Q_EXTP);
    if (threadIdx.x == 0) {
        ctl.closest_rays += L.closest_rays;
        if (COUNT) { st.stats[blockIdx.x].nodes_visited += L.nodes; st.stats[blockIdx.x].prims_tested += L.prims; }
    }
}

// stand-alone batched closest-hit query (tghip_trace_rays) on caller rays
// WIDE: the scene has a wide BVH and no instances (the walk the wavefront kernels do, one ray per lane without refills)
template<bool COUNT, bool FLAT, int INST = 0, bool WIDE = false>
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
        if (WIDE) hits[i] = traverseClosestWide<COUNT, KINDS_ALL, INST != 0>(s, ray, reinterpret_cast<uint2 *>(ldsStack) + threadIdx.x, blockDim.x, nodes, prims, hitInst);
        else hits[i] = INST ? traverseClosestInst<COUNT, INST == 2 ? KINDS_MESH : KINDS_ALL>(s, ray, ldsStack + threadIdx.x, blockDim.x, nodes, prims, hitInst)
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
//                and forwards hits of the classes 1 .. 3 (hit record stored) to their shading queues;
//   FUSE_SHADOW  the <= 2 shadow rays of a vertex are any-hit tested inline instead of being queued for
//                k_trace_shadow, so the NEE term is added on the spot and no shadow record is written.
//   FUSE_LOOP    (with both of the above, scenes whose materials are all of class 0) the workgroup runs its slots to completion
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
// (the kernel's body as a function of the workgroup's LDS objects: k_shade below and k_tail run it; returns whether the workgroup's
// extension queues hold work -- the FUSE launches report it)
// STAGED: sg's small tables are in LDS already (k_tail stages them once for all its iterations)
// GLOBAL_TABLES: the scene's small tables do not fit the LDS copy (stageSceneTables): read them where they are
template<uint32_t M, int FUSE, bool STAGED = false, bool GLOBAL_TABLES = false>
PT_DEV bool shadeBody(const DeviceScene &sg, const PathState &st, const PassParams &pp, int cls, BlockLds &L, unsigned char *ldsTables, unsigned short *order)
{
    // (the fused flat-list launches and k_tail -- STAGED -- re-read their slots within microseconds: no non-temporal hint there, pt_kernels.h)
    constexpr int SNT = (FUSE != 0 || (STAGED && !PT_NT_TAIL)) ? 0 : PT_NT_STATE;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    // (CLS_MISS: the escaped paths.  CLS_0_AND_MISS: class 0, then the escaped paths -- one launch of the variant both run; the expanded list
    // keeps the two runs apart, so at most one wave per workgroup mixes surface shading with escaped paths)
    const int qIn = (FUSE & FUSE_TRACE) ? Q_EXTP : cls == CLS_0_AND_MISS ? Q_SHADE0 : shadeQueue(cls);
    const int qIn2 = (FUSE & FUSE_TRACE) ? Q_EXT : cls == CLS_0_AND_MISS ? Q_MISS : -1;
    const uint32_t appendMask = (1u << Q_EXT) | (1u << Q_EXTP) | (1u << Q_SHADOW) | ((FUSE & FUSE_TRACE) ? ((1u << Q_SHADE1) | (1u << Q_SHADE2) | (1u << Q_SHADE3)) : 0u) | (FUSE == 0 ? (1u << Q_FIN) : 0u);
    // the wavefront launches (FUSE == 0) of the shading classes of one iteration run concurrently (runBatch): each consumes its own
    // queue, touches its own slots, and ORs what it appends into the workgroup's global bitmaps
    constexpr bool CONCURRENT = FUSE == 0;
    queuesBegin(L, st, ctl, qIn, appendMask, order, qIn2, CONCURRENT);
    if (CONCURRENT && L.n == 0u)
        return false;                            // nothing of this class in the workgroup: no bitmap changes, nothing to write back
    const DeviceScene s = (STAGED || GLOBAL_TABLES) ? sg : stageSceneTables(sg, ldsTables);
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const int maxBounces = s.settings.max_bounces, minBounces = s.settings.min_bounces;
    // PT_EXP_HALF (a compile-only diagnostic, never the product): 1 = the kernel without the continuation sample, 2 = without next-event
    // estimation -- the register demand of the two halves a split shading stage would consist of (profiles/r5_shade_split_halves.txt)
#if defined(PT_EXP_HALF) && PT_EXP_HALF == 2
    const bool nee = false;
#else
    const bool nee = s.settings.enable_light_sampling != 0;
#endif
    uint32_t finishedCount = 0, fusedClosest = 0, fusedShadow = 0, fusedPrims = 0, fusedNodes = 0;
    PROF_DECL;

    // FUSE_LOOP runs without queues: a finished path is regenerated in place, so a slot stays busy until the workgroup's
    // work items run out.  Thread t owns slots t, t + blockDim, ... for the whole launch (consecutive lanes = consecutive
    // slots: every state access is a full cache line), `idle` has one bit per owned slot that has nothing left to do,
    // and every wave leaves the loop on its own -- no barrier, no bitmap traffic between the wavefront iterations.
    constexpr bool DIRECT = (FUSE & FUSE_LOOP) != 0;
    uint32_t idle = 0;
    // Round 6 experiment (PT_TRACE_AHEAD = 1, not the product): the one-launch render traces a slot's NEXT ray -- the continuation, or the camera ray of the
    // path that takes the slot over -- at the END of the turn, and a continuation that leaves the scene ends its path right there.  Every path's last visit
    // is a miss (its ray leaves the Cornell box through the open front; no environment to ask): with 3.9 vertices per path a quarter of a turn's lanes hold
    // such a path, find that out by tracing, and sit out the shading sections -- 47 of 64 lanes at any spp.  With this a visit starts from a stored hit
    // (53 lanes: the camera rays beside the box still miss) and the render needs 11 % fewer turns -- of 20.8 us instead of 16.2: a walk of the flat list
    // costs the WAVE its ~4 us whether 48 or 17 lanes need it, and there are two of them per turn now.  Cornell box 2 545 -> 2 290 Msamples/s; images identical
    // (the GPU suite passes on it).  profiles/r6_ab_trace_ahead.txt.  Only where a miss has nothing else to do: no infinite lights, media, auxiliary outputs.
    constexpr bool AHEAD = PT_TRACE_AHEAD && DIRECT && (FUSE & FUSE_TRACE) != 0 && (M & (FEAT_INFINITE | FEAT_MEDIA | FEAT_AUX)) == 0u;
    uint32_t traced = 0;                         // AHEAD: one bit per owned slot whose A_HIT holds the hit of the ray in the slot
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
    const bool aborted = __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (uint32_t base = 0, turn = 0; base < n; base += blockDim.x, ++turn) {
        uint32_t i = base + threadIdx.x;
        PROF(0);
        bool hasShadow = false, finished = false, survives = false, black = false, toComplex = false;
        int complexCls = 1;                      // FUSE_TRACE: the class whose launch shades the hit this launch forwards
        uint32_t slot = 0, local = 0;
        f3 em = splat3(0.0f);
        if (DIRECT ? !((idle >> turn) & 1u) : i < n) {
            f3 pendingOut = splat3(0.0f);
            local = DIRECT ? i : order[i];
            slot = first + local;
            float4 ro = slotF4<SNT>(st, A_RAY_O, slot), rd = slotF4<SNT>(st, A_RAY_D, slot), hit, thr4 = slotF4<SNT>(st, A_THR, slot);
            if (AHEAD && ((traced >> turn) & 1u)) {
                hit = slotF4<SNT>(st, A_HIT, slot);                   // traced at the end of the slot's last visit (class 0 throughout: FUSE_LOOP)
            } else if (FUSE & FUSE_TRACE) {
                // TraceableScene::intersect inline: the flat record list, walked uniformly by the wave
                RayD r0;
                r0.o = xyz(ro); r0.d = xyz(rd); r0.tmin = ro.w; r0.tmax = rd.w;
                hit = traverseClosest<true, true, shadeKinds(M)>(sg, r0, nullptr, 0, fusedNodes, fusedPrims);
                fusedClosest++;
                int ri = __float_as_int(hit.w);
                complexCls = ri >= 0 ? (int)at32(sg.rec_class, (uint32_t)ri) : 0;
                toComplex = complexCls != 0;
                if (toComplex)
                    slotF4<SNT>(st, A_HIT, slot) = hit;           // shaded by its class's launch, which follows
            } else {
                hit = slotF4<SNT>(st, A_HIT, slot);
            }
          if (!toComplex) {
            float4 em4 = slotF4<SNT>(st, A_EMI, slot);
            em = xyz(em4);
            const int hitInst = ((M & FEAT_INSTANCES) && s.num_instances) ? __float_as_int(em4.w) : -1;   // written by k_trace_closest<.., INST>
            uint4 misc = slotU4<SNT>(st, A_MISC, slot);
            uint2 rs = make_uint2(misc.x, misc.y);
            uint32_t pixel = misc.z;
            Rng rng;
            rng.state = ((uint64_t)rs.y << 32) | rs.x;
            rng.inc = ((uint64_t)pixel << 1) | 1u;
            rng.sobol = nullptr;
            rng.scramble = rng.index = rng.dim = 0u;
            if ((M & FEAT_QMC) && (pp.flags & TGHIP_PASS_SOBOL)) {
                uint4 sm = slotU4<SNT>(st, A_SAMP, slot);             // .x = sample index, .z = next dimension
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
                if (auxOn && !recorded) { aux0 = slotF4<SNT>(st, A_AUX0, slot); aux1 = slotF4<SNT>(st, A_AUX1, slot); }
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
                    if (!mediumSampleDistance<M>(s, med, rng, ray.o, ray.d, __float_as_int(hit.w) >= 0 ? hit.x : PT_INF, medBounce, w, t, exited)) {
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
                        float mis0 = 1.0f, mis1 = 1.0f;       // (media scenes always run with st.nee_factors: the factors stay apart, A_NEE0 .. A_NEE2)
                        const bool meshLight = s.objects[light].type == TGHIP_OBJ_MESH;
                        const bool diracLight = s.objects[light].type == TGHIP_OBJ_POINT;
                        {
                            f3 d; float dist, pdf;
                            if (lightSampleDirect<M>(s, light, volP, rng, d, dist, pdf)) {
                                float f = phaseEval(mm, ray.d, d);
                                if (f != 0.0f && meshLight) {
                                    mis0 = powerHeuristic(pdf, f);                         // phase pdf == phase value
                                    slotF4<SNT>(st, A_SH_D0, slot) = mk4(d, dist);
                                    slotF4<SNT>(st, A_SH_C0, slot) = mk4(splat3(f), __uint_as_float(tag));
                                    slotF4<SNT>(st, A_NEE0, slot) = make_float4(0.0f, 0.0f, 0.0f, pdf);
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
                                            if (!diracLight)
                                                mis0 = powerHeuristic(pdf, f);
                                            slotF4<SNT>(st, A_SH_D0, slot) = mk4(d, lh.t);
                                            slotF4<SNT>(st, A_SH_C0, slot) = mk4(splat3(f), __uint_as_float(tag));
                                            slotF4<SNT>(st, A_NEE0, slot) = mk4(e, pdf);
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
                                slotF4<SNT>(st, A_SH_D1, slot) = mk4(w, ppdf);                      // directPdf needs the hit
                                slotF4<SNT>(st, A_SH_C1, slot) = mk4(splat3(1.0f), __uint_as_float(tag));
                                q1 = true;
                            } else {
                                RayD sr; sr.o = volP; sr.d = w; sr.tmin = 0.0f; sr.tmax = PT_INF;
                                LightHit lh;
                                if (lightIntersect<M>(s, light, sr, lh)) {
                                    f3 e = lightEvalDirect<M>(s, light, lh.u, lh.v, lh.backSide);
                                    if (!isZero(e)) {
                                        mis1 = powerHeuristic(ppdf, lightDirectPdf<M>(s, light, w, volP, lh));
                                        slotF4<SNT>(st, A_SH_D1, slot) = mk4(w, lh.t);
                                        slotF4<SNT>(st, A_SH_C1, slot) = mk4(splat3(1.0f), __uint_as_float(tag));   // phaseSample.weight
                                        slotF4<SNT>(st, A_NEE1, slot) = mk4(e, 0.0f);
                                        q1 = true;
                                    }
                                }
                            }
                        }
                        if (q0 || q1) {
                            hasShadow = true;
                            if (!q0) slotF4<SNT>(st, A_SH_C0, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                            if (!q1) slotF4<SNT>(st, A_SH_C1, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                            slotF4<SNT>(st, A_SH_O, slot) = mk4(volP, 0.0f);
                            slotF4<SNT>(st, A_SH_W, slot) = mk4(throughput, lightWeight);
                            slotF4<SNT>(st, A_NEE2, slot) = make_float4(mis0, mis1, 0.0f, 0.0f);
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
                Frame frame = shadingFrame<M>(s, info);
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
                            // st.nee_factors: the two terms' factors stay apart for the shadow kernel (A_NEE0 .. A_NEE2, pt_kernels.h)
                            const bool factors = !(FUSE & FUSE_SHADOW) && st.nee_factors != 0u;
                            float mis0 = 1.0f, mis1 = 1.0f;
                            const bool meshLight = (M & FEAT_MESHLIGHT) && s.objects[light].type == TGHIP_OBJ_MESH;
                            const bool diracLight = (M & FEAT_SOLIDS) && s.objects[light].type == TGHIP_OBJ_POINT;
                            // lightSample (TraceBase.cpp:246-285)
                            {
                                f3 d; float dist, pdf;
                                const bool sampledLight = lightSampleDirect<M>(s, light, info.p, rng, d, dist, pdf);
                                PROF(9);
                                if (sampledLight) {
                                    if (M & FEAT_MEDIA) tag = mediaTag(d);
                                    ev.wo = toLocal(frame, d);
                                    ev.requested = LOBE_ALL_BUT_SPECULAR;
                                    if (isConsistent(ev.wo, d)) {
                                        f3 f = bsdfEval<M>(s, info.bsdf, ev);
                                        if ((M & FEAT_MESHLIGHT) && !isZero(f) && meshLight) {
                                            // mesh emitter: whether the ray reaches the light, and with which emission, is only
                                            // known after the scene traversal (TriangleMesh::intersect is a BVH query), so the
                                            // shadow kernel completes f*e/pdf * powerHeuristic from these factors
                                            // (a scene with a mesh emitter always runs with nee_factors)
                                            mis0 = powerHeuristic(pdf, bsdfPdf<M>(s, info.bsdf, ev));
                                            slotF4<SNT>(st, A_SH_D0, slot) = mk4(d, dist);
                                            slotF4<SNT>(st, A_SH_C0, slot) = mk4(f, __uint_as_float(tag));
                                            slotF4<SNT>(st, A_NEE0, slot) = make_float4(0.0f, 0.0f, 0.0f, pdf);
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
                                                    if (!diracLight) {                            // no MIS against a Dirac light (:281-282)
                                                        mis0 = powerHeuristic(pdf, bsdfPdf<M>(s, info.bsdf, ev));
                                                        lightF = lightF*mis0;
                                                    }
                                                    if (factors) {
                                                        slotF4<SNT>(st, A_SH_D0, slot) = mk4(d, lh.t);
                                                        slotF4<SNT>(st, A_SH_C0, slot) = mk4(f, __uint_as_float(tag));
                                                        slotF4<SNT>(st, A_NEE0, slot) = mk4(e, pdf);
                                                    } else if (FUSE & FUSE_SHADOW) {
                                                        sr.tmax = lh.t;
                                                        fusedShadow++;
                                                        if (!traverseOccluded<true, true, shadeKinds(M)>(sg, sr, light, nullptr, 0, fusedNodes, fusedPrims) && bounce + 1 >= minBounces)
                                                            inlineResult = inlineResult + lightF;
                                                    } else {
                                                        slotF4<SNT>(st, A_SH_D0, slot) = mk4(d, lh.t);
                                                        slotF4<SNT>(st, A_SH_C0, slot) = mk4(lightF, __uint_as_float(tag));
                                                    }
                                                    q0 = true;
                                                }
                                            }
                                        }
                                    }
                                }
                            }
                            PROF(12);
                            // bsdfSample (TraceBase.cpp:287-321); not for Dirac lights (:396-397)
                            if (!diracLight) {
                                ev.requested = LOBE_ALL_BUT_SPECULAR;
                                ev.weight = splat3(1.0f); ev.pdf = 1.0f;
                                if (bsdfSample<M>(s, info.bsdf, ev) && !isZero(ev.weight)) {
                                    f3 wog = toGlobal(frame, ev.wo);
                                    if (M & FEAT_MEDIA) tag = mediaTag(wog);
                                    if ((M & FEAT_MESHLIGHT) && meshLight) {
                                        if (isConsistent(ev.wo, wog)) {
                                            slotF4<SNT>(st, A_SH_D1, slot) = mk4(wog, ev.pdf);          // directPdf needs the hit
                                            slotF4<SNT>(st, A_SH_C1, slot) = mk4(ev.weight, __uint_as_float(tag));
                                            q1 = true;
                                        }
                                    } else if (isConsistent(ev.wo, wog)) {
                                        RayD sr; sr.o = info.p; sr.d = wog; sr.tmin = 5e-4f; sr.tmax = PT_INF;
                                        LightHit lh;
                                        if (lightIntersect<M>(s, light, sr, lh)) {
                                            // the emission lookup and the light's pdf lookup are issued together (both are dependent
                                            // loads of a bitmap light); the pdf is only USED behind the !isZero(e) test, as in :312-316
                                            f3 e = lightEvalDirect<M>(s, light, lh.u, lh.v, lh.backSide);
                                            const float lightPdf = lightDirectPdf<M>(s, light, wog, info.p, lh);
                                            if (!isZero(e)) {
                                                f3 bsdfF = e*ev.weight;
                                                mis1 = powerHeuristic(ev.pdf, lightPdf);
                                                bsdfF = bsdfF*mis1;
                                                if (factors) {
                                                    slotF4<SNT>(st, A_SH_D1, slot) = mk4(wog, lh.t);
                                                    slotF4<SNT>(st, A_SH_C1, slot) = mk4(ev.weight, __uint_as_float(tag));
                                                    slotF4<SNT>(st, A_NEE1, slot) = mk4(e, 0.0f);
                                                } else if (FUSE & FUSE_SHADOW) {
                                                    sr.tmax = lh.t;
                                                    fusedShadow++;
                                                    if (!traverseOccluded<true, true, shadeKinds(M)>(sg, sr, light, nullptr, 0, fusedNodes, fusedPrims) && bounce + 1 >= minBounces)
                                                        inlineResult = inlineResult + bsdfF;
                                                } else {
                                                    slotF4<SNT>(st, A_SH_D1, slot) = mk4(wog, lh.t);
                                                    slotF4<SNT>(st, A_SH_C1, slot) = mk4(bsdfF, __uint_as_float(tag));
                                                }
                                                q1 = true;
                                            }
                                        }
                                    }
                                }
                            }
                            PROF(13);
                            if ((FUSE & FUSE_SHADOW) && (q0 || q1)) {
                                // emission += estimateDirect(...)*throughput, like k_trace_shadow
                                em = em + (inlineResult*lightWeight)*throughput;
                            } else if (q0 || q1) {
                                hasShadow = true;
                                auxVisPending = q0;
                                if (!q0) slotF4<SNT>(st, A_SH_C0, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                                if (!q1) slotF4<SNT>(st, A_SH_C1, slot) = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
                                slotF4<SNT>(st, A_SH_O, slot) = mk4(info.p, 5e-4f);
                                slotF4<SNT>(st, A_SH_W, slot) = mk4(throughput, lightWeight);
                                if (factors) slotF4<SNT>(st, A_NEE2, slot) = make_float4(mis0, mis1, 0.0f, 0.0f);
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
#if defined(PT_EXP_HALF) && PT_EXP_HALF == 1
                    if (true) {
#else
                    if (!bsdfSample<M>(s, info.bsdf, ev)) {
#endif
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
            PROF(14);
            if (state == ST_ACTIVE) {
                slotF4<SNT>(st, A_RAY_O, slot) = mk4(ray.o, ray.tmin);
                slotF4<SNT>(st, A_RAY_D, slot) = mk4(ray.d, ray.tmax);
                *reinterpret_cast<uint2 *>(&slotUW(st, A_MISC, slot, 0u)) = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                if ((M & FEAT_QMC) && (pp.flags & TGHIP_PASS_SOBOL))
                    slotUW(st, A_SAMP, slot, 2u) = rng.dim;
            }
            if constexpr ((M & FEAT_AUX) != 0u) if (auxOn) {
                if (!recorded && state != ST_ACTIVE) {       // the sample returned early: it adds nothing (hitDistance is not a depth)
                    aux0.w = __uint_as_float(0x7FC00000u);
                    auxStore = true;
                }
                if (auxStore) { slotF4<SNT>(st, A_AUX0, slot) = aux0; slotF4<SNT>(st, A_AUX1, slot) = aux1; }
            }
            const uint32_t newFlags = FLAG_MAKE(bounce, wasSpecular, state) | ((M & FEAT_MEDIA) ? FLAG_MEDIUM_BITS(med, medBounce) : 0u)
                                    | (recorded ? FLAG_AUX_RECORDED : 0u);
            survives = state == ST_ACTIVE;
            black = state == ST_TERMINATED_BLACK;
            if (hasShadow) {
                // k_trace_shadow adds the NEE term, then finishes the path if it ended here
                slotF4<SNT>(st, A_EMI, slot) = mk4(em, 0.0f);
                slotF4<SNT>(st, A_SH_P, slot) = mk4(pendingOut, __uint_as_float(newFlags));
            } else if (survives) {
                slotF4<SNT>(st, A_EMI, slot) = mk4(em, 0.0f);
            } else {
                finished = true;
                if (FUSE == 0) {                 // k_finish finalises the sample and regenerates the slot (below)
                    slotF4<SNT>(st, A_EMI, slot) = mk4(em, 0.0f);
                    slotW(st, A_SH_P, slot, 3u) = __uint_as_float(newFlags);
                }
            }
            if (survives)
                slotF4<SNT>(st, A_THR, slot) = mk4(throughput, __uint_as_float(newFlags));
            if constexpr (AHEAD) {
                bool haveHit = false;
                if (survives) {
                    const float4 h = traverseClosest<true, true, shadeKinds(M)>(sg, ray, nullptr, 0, fusedNodes, fusedPrims);
                    fusedClosest++;
                    if (__float_as_int(h.w) < 0) {
                        // the continuation leaves the scene: what the escaped branch above would do with it at the slot's next visit
                        survives = false;
                        finished = true;
                        black = isnan(sum3(throughput) + sum3(em));
                    } else {
                        slotF4<SNT>(st, A_HIT, slot) = h;
                        haveHit = true;
                    }
                }
                traced = haveHit ? (traced | (1u << turn)) : (traced & ~(1u << turn));
            }
            PROF(15);
          }
        }
        PROF(5);
        if (!DIRECT) {
            if (FUSE & FUSE_TRACE) queuePush(toComplex, local, L, shadeQueue(complexCls));
            queuePush(hasShadow, local, L, Q_SHADOW);
        }
        PROF(6);
        // Wavefront launches (FUSE == 0) leave finalising a finished sample and starting the slot's next camera path to k_finish, the
        // last launch of the iteration, which does so for the paths that ended at their shadow rays anyway: nextPath's camera, filter
        // and work-item code stays out of the shading variants' register budget, and in k_finish every lane regenerates instead of
        // the few of a shading wave whose path happened to end.  The fused flat-list launches regenerate in place.
        bool regenerated = false;
        if constexpr (FUSE != 0)
            regenerated = nextPath<true, (M & FEAT_QMC) != 0, SNT>(s, st, pp, finished, false, slot, em, black, &L.cursor, aborted, finishedCount);
        if constexpr (AHEAD) {
            if (regenerated) {                   // the camera ray nextPath just wrote: traced now, its hit (or miss: the next visit ends the path) stored
                const float4 ro = slotF4<SNT>(st, A_RAY_O, slot), rd = slotF4<SNT>(st, A_RAY_D, slot);
                RayD r1;
                r1.o = xyz(ro); r1.d = xyz(rd); r1.tmin = ro.w; r1.tmax = rd.w;
                const float4 h = traverseClosest<true, true, shadeKinds(M)>(sg, r1, nullptr, 0, fusedNodes, fusedPrims);
                fusedClosest++;
                slotF4<SNT>(st, A_HIT, slot) = h;
                traced |= 1u << turn;
            }
        }
        PROF(7);
        if (DIRECT) {
            if (finished && !regenerated) idle |= 1u << turn;   // the work items ran out: nothing left for this slot
        } else {
            queuePush(survives, local, L, Q_EXT);
            if (FUSE != 0) queuePush(regenerated, local, L, Q_EXTP);
            else           queuePush(finished, local, L, Q_FIN);
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
    const bool anyExt = queuesEnd(L, st, qIn, appendMask, qIn2, CONCURRENT);
    if (FUSE != 0 && threadIdx.x == 0) {         // (the wavefront launches neither regenerate nor finish samples: k_finish does)
        ctl.item_cursor = L.cursor;
        ctl.samples += L.samples;
        if (FUSE) {
            ctl.closest_rays += L.closest_rays; ctl.shadow_rays += L.shadow_rays;
            st.stats[blockIdx.x].prims_tested += L.prims;
            // fused mode has no k_trace_shadow launch: the last shading launch of the iteration reports liveness
            if (anyExt) atomicMax(&st.live[0], (uint32_t)pp.iter_tag);
        }
    }
    return anyExt;
}
template<uint32_t M, int W, int FUSE, bool GLOBAL_TABLES = false>
__global__ __launch_bounds__(256, W) void k_shade(DeviceScene sg, PathState st, PassParams pp, int cls)
{
    __shared__ BlockLds L;
    __shared__ __attribute__((aligned(16))) unsigned char ldsTables[GLOBAL_TABLES ? 16u : PT_LDS_TABLE_BYTES];
    __shared__ unsigned short order[PT_MAX_SLOTS_PER_BLOCK];
    (void)shadeBody<M, FUSE, false, GLOBAL_TABLES>(sg, st, pp, cls, L, ldsTables, order);
}

// TraceBase::generalizedShadowRay (TraceBase.cpp:62-125) for the shadow rays queued by k_shade:
// a closest-hit query up to the light; unoccluded iff nothing is hit or the closest hit is the
// light itself (endCap); surfaces with a forward lobe attenuate and the ray continues (FORWARD variant only:
// scenes without a forward-lobe BSDF run the lean variant).
template<bool COUNT, bool FORWARD, bool FLAT, int INST = 0>
__global__ __launch_bounds__(512) void k_trace_shadow(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    extern __shared__ int ldsStack[];
    __shared__ BlockLdsSmall L;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    queuesBegin(L, st, ctl, Q_SHADOW, (1u << Q_EXT) | (1u << Q_EXTP), reinterpret_cast<unsigned short *>(ldsStack));
    const uint32_t n = L.n;
    const OrderRegs ord = orderPreload(reinterpret_cast<unsigned short *>(ldsStack), n);
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const bool aborted = __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
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
                f3 meshE = splat3(0.0f);                       // mesh emitters: the emission the walk found at the light ...
                float meshMis = 1.0f;                          // ... and, for the bsdf ray, its power-heuristic weight
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
                    // The quad light the ray is aimed at is only FOUND when Embree's slab test lets the ray into its flat box (oracle.c:
                    // generalizedShadowRay): culled, the function returns as at the end cap, but ray.farT() was not shortened to the hit
                    bool found = ri >= 0;
                    if (found && hitObject == endCap && s.objects[endCap].type == TGHIP_OBJ_QUAD) {
                        const TgHipObject &lo = s.objects[endCap];
                        const f3 b = ld3(lo.base), e0 = ld3(lo.edge0), e1 = ld3(lo.edge1);
                        const f3 p1 = b + e0, p2 = b + e1, p3 = (b + e0) + e1;
                        const f3 bl = mk3(fminf(fminf(b.x, p1.x), fminf(p2.x, p3.x)), fminf(fminf(b.y, p1.y), fminf(p2.y, p3.y)), fminf(fminf(b.z, p1.z), fminf(p2.z, p3.z)));
                        const f3 bh = mk3(fmaxf(fmaxf(b.x, p1.x), fmaxf(p2.x, p3.x)), fmaxf(fmaxf(b.y, p1.y), fmaxf(p2.y, p3.y)), fmaxf(fmaxf(b.z, p1.z), fmaxf(p2.z, p3.z)));
                        found = embreeBoxVisible(ray.o, ray.d, ray.tmin, ray.tmax, bl, bh);
                    }
                    if (medium >= 0)                             // TraceBase.cpp:103-112: ray.farT() is the hit distance when anything was hit
                        transmittance = transmittance*mediumTransmittance(s, medium, ray.o, ray.d, found ? hit.x : ray.tmax, startsOnSurface, true);
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
                            } else {
                                float directPdf = lengthSq(xyz(so) - li.p)/(-dot(ray.d, li.Ng)*lo.area);
                                meshMis = powerHeuristic(sd.w, directPdf);
                            }
                            meshE = e;
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
                // The term in the reference's order of operations (st.nee_factors: the shading kernel left the factors apart): attenuatedEmission
                // returns shadow*light.evalDirect (TraceBase.cpp:173), lightSample (f*e)/pdf, then *= the power heuristic unless the light is a
                // Dirac one (:277-282; its weight is stored as 1), bsdfSample (e*weight), then *= the power heuristic (:316-318); the volume
                // pair likewise (:346-351, 374-378).
                const bool dark = isZero(transmittance);       // `if (shadow == 0.0f) return Vec3f(0.0f)` (:170-171)
                const float4 n0 = slotF4(st, A_NEE0, slot), n1 = slotF4(st, A_NEE1, slot), mis = slotF4(st, A_NEE2, slot);
                const f3 e = transmittance*(meshLight ? meshE : r == 0 ? xyz(n0) : xyz(n1));
                if (r == 0) transmittance = (xyz(c)*e)/n0.w*mis.x;
                else        transmittance = (e*xyz(c))*(meshLight ? meshMis : mis.y);
                if (dark || isZero(e)) transmittance = splat3(0.0f);   // `if (e == 0.0f) return Vec3f(0.0f)` (:273-274, 311-312)
                }
                if (!FORWARD) shadowT = transmittance;
                if (r == 0 && (pp.flags & TGHIP_PASS_AUX)) {   // the visibility output of the vertex that recorded (PathTracer.cpp:93-94)
                    float &a1w = slotW(st, A_AUX1, slot, 3u);
                    if (isinf(a1w))
                        a1w = visValid ? avg3(shadowT) : __uint_as_float(0x7FC00000u);
                }
                if (FORWARD)                     result = result + transmittance;        // (the whole term, above)
                else if (!isZero(transmittance)) result = result + xyz(c)*transmittance;
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
        if (anyExt) atomicMax(&st.live[0], iterTag);   // (max: the parts of the pool run on streams of their own and pass here out of order)
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
    __shared__ BlockLdsSmall L;
    __shared__ uint32_t fetchNext;
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    int *stack = ldsDyn + (st.slots_per_block >> 1) + threadIdx.x;
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
                float4 n0 = ld4(nd, 0u), n1 = ld4(nd, 1u), n2 = ld4(nd, 2u), n3 = ld4(nd, 3u);
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

// k_trace_shadow_dyn over the 8-wide BVH: any-hit queries, one memory round trip (a node or a record) per lane and loop
// turn, like k_trace_closest_wide.  Dynamic LDS: [expanded queue, 2 B per slot][group stacks, wideDepth x 8 B per thread].
// JOIN = false (INST only) leaves PT_TURN_JOIN out: the miscompiled variant, kept for tools/repro_latch_miscompile.py
template<bool COUNT, bool SOLIDS = true, bool INST = false, bool JOIN = true>
__global__ WIDE_SHADOW_BOUNDS void k_trace_shadow_wide(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    uint2 *stack = reinterpret_cast<uint2 *>(ldsDyn + (st.slots_per_block >> 1)) + threadIdx.x;
    const int stride = (int)blockDim.x;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) fetchNext = 0;
    const unsigned long long wpStart = COUNT ? wall_clock64() : 0ull;
    uint32_t turns = 0, dryTurns = 0;
    // (the extension and hold queues are loaded and written back too: suspending / resolving a slot moves its path's bit between them)
    const uint32_t appendMask = (1u << Q_FIN) | (!INST && st.suspend_lanes != 0u ? (1u << Q_EXT) | (1u << Q_HOLD) : 0u);
    queuesBegin(L, st, ctl, Q_SHADOW, appendMask, order);
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
    WideRay wr; wr.idir = splat3(1.0f); wr.octInv = 0u;
    WideState w;
    wideStart(w);
    f3 contrib = splat3(0.0f);
    int endCap = -1;
    bool exhausted = false;
    uint32_t age = 0;                            // turns this lane's slot has had in this launch (PathState::suspend_turns)
    bool held = false;                           // the slot is a resumed one: its path's extension ray may sit in Q_HOLD
    const bool maySuspend = !INST && st.suspend_lanes != 0u && n >= st.suspend_min_queue;

    // sets up ray `r` (or the next valid one) of the current slot; returns false when the slot has no ray left
    // (resume: ray `r` itself, whose suspended walk -- already in `w` -- goes on: the checks below passed when it was first set up)
    auto setupRay = [&](bool resume = false) -> bool {
        for (; r < 2; ++r) {
            float4 c = r == 0 ? slotF4(st, A_SH_C0, slot) : slotF4(st, A_SH_C1, slot);
            uint32_t tag = __float_as_uint(c.w);
            if (tag == 0xFFFFFFFFu && !resume)
                continue;
            float4 sd = r == 0 ? slotF4(st, A_SH_D0, slot) : slotF4(st, A_SH_D1, slot);
            endCap = (int)(tag & 0xFFFFFFu);
            int bounce = (int)(tag >> 24);
            if (!resume) rays++;
            if (bounce < minBounces && !resume)
                continue;                        // contributes nothing (TraceBase.cpp:114-115 with minBounces)
            contrib = xyz(c);
            ray.o = so; ray.d = xyz(sd); ray.tmin = eps; ray.tmax = sd.w;
            wr = wideRaySetup(ray);
            if (!resume) wideStart(w);
            return true;
        }
        return false;
    };
    // NEE term -> path radiance; paths that ended at this vertex go on the finished list
    auto finishSlot = [&]() {
        float4 wgt = slotF4(st, A_SH_W, slot);
        float4 p = slotF4(st, A_SH_P, slot);
        f3 em = xyz(slotF4(st, A_EMI, slot));
        em = em + (result*wgt.w)*xyz(wgt);       // emission += estimateDirect(...)*throughput
        em = em + xyz(p);
        slotF4(st, A_EMI, slot) = mk4(em, 0.0f);
        queuePush(FLAG_STATE(__float_as_uint(p.w)) != ST_ACTIVE, local, L, Q_FIN);
        if constexpr (!INST) {
            if (held) {                          // the path's extension ray was held back while this slot was suspended: release it
                const uint32_t bit = 1u << (local & 31u);
                if (L.bm[Q_HOLD][local >> 5] & bit) {
                    atomicAnd(&L.bm[Q_HOLD][local >> 5], ~bit);
                    atomicOr(&L.bm[Q_EXT][local >> 5], bit);
                }
            }
        }
        busy = false;
    };

    WALK_PROF_DECL;
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
                    float4 o4 = slotF4(st, A_SH_O, slot);
                    so = xyz(o4); eps = o4.w;
                    bool resumed = false;
                    if constexpr (!INST) {
                        resumed = (__float_as_uint(o4.w) & WALK_SUSPENDED_BIT) != 0u;   // a slot an earlier launch suspended (below)
                        eps = __uint_as_float(__float_as_uint(o4.w) & ~WALK_SUSPENDED_BIT);
                    }
                    held = resumed;
                    age = 0;
                    busy = true;
                    if (!resumed) {
                        slots++;
                        result = splat3(0.0f);
                        r = 0;
                        if (!setupRay())
                            finishSlot();
                    } else {
                        walkRestore(st, slot, w, stack, stride);
                        if (COUNT) wpResumed++;
                        const float4 part = slotF4(st, st.walk_base + 2u, slot);
                        result = xyz(part); r = __float_as_int(part.w);
                        slotW(st, A_SH_O, slot, 3u) = eps;                   // (an ordinary shadow slot again)
                        (void)setupRay(true);
                    }
                }
            }
            if (base + (uint32_t)__popcll(want) >= n) {
                exhausted = true;
                if (COUNT) wpDry = wall_clock64();
            }
            busyMask = __ballot(busy);
        }
        if constexpr (!INST) {
            // the queue is dry and the wave is down to its last, longest walks: suspend those that have had their turns (k_trace_closest_wide).
            // The slot stays on the shadow queue; its path's extension ray, if it has one, waits in Q_HOLD until the slot is resolved.
            if (maySuspend && exhausted && (uint32_t)__popcll(busyMask) <= st.suspend_lanes) {
                if (busy && age >= st.suspend_turns) {
                    walkSave(st, slot, w, stack, stride);
                    if (COUNT) wpSuspended++;
                    slotF4(st, st.walk_base + 2u, slot) = mk4(result, __int_as_float(r));
                    slotW(st, A_SH_O, slot, 3u) = __uint_as_float(__float_as_uint(eps) | WALK_SUSPENDED_BIT);
                    queuePush(true, local, L, Q_SHADOW);
                    const uint32_t bit = 1u << (local & 31u);
                    if (L.bm[Q_EXT][local >> 5] & bit) {
                        atomicAnd(&L.bm[Q_EXT][local >> 5], ~bit);
                        atomicOr(&L.bm[Q_HOLD][local >> 5], bit);
                    }
                    busy = false;
                }
                busyMask = __ballot(busy);
            }
        }
        if (busyMask == 0ull)
            break;
        age++;
        if (COUNT) { turns++; dryTurns += exhausted ? 1u : 0u; if (exhausted) wpBusyDry += (uint32_t)__popcll(busyMask); else wpBusy += (uint32_t)__popcll(busyMask); }
        if constexpr (!INST) {
            // a record that is the last one of its node is fetched together with the node the walk visits next (k_trace_closest_wide)
            if (busy) {
                uint32_t recIdx = 0, nodeIdx = 0, idx = 0;
                bool hasRec = false, hasNode = false, walkOver = false;
                const int what = wideNext<false>(w, wr.octInv, stack, stride, idx);
                if (what == 1) {
                    hasRec = true; recIdx = idx;
                    if (w.triMask == 0u && st.leaf_batch != 9u) {
                        const int next = wideNext<false>(w, wr.octInv, stack, stride, idx);
                        if (next == 2) { hasNode = true; nodeIdx = idx; } else walkOver = true;
                    }
                } else if (what == 2) { hasNode = true; nodeIdx = idx; }
                else walkOver = true;
                float4 r0, r1, r2;
            WideNodeRegs nd;
                if (hasRec) { r0 = at32(s.recs, recIdx*3u + 0u); r1 = at32(s.recs, recIdx*3u + 1u); r2 = at32(s.recs, recIdx*3u + 2u); }
                if (hasNode) wideNodeFetch(nd, reinterpret_cast<const char *>(s.wide), wideNodeOff(s, nodeIdx), wr);
                bool rayDone = false;
                if (hasRec) {
                    if (COUNT) prims++;
                    float tmax = ray.tmax;
                    float4 hit;
                    uint32_t meta;
                    if (testRecordLoaded<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, recIdx, r0, r1, r2, ray, tmax, hit, meta) && (int)TGHIP_REC_OBJECT(meta) != endCap)
                        rayDone = true;          // occluded: the node fetched alongside is not visited
                }
                if (!rayDone && hasNode) {
                    if (COUNT) nodes++;
                    wideVisit(w, nd, ray.o, wr, ray.tmin, ray.tmax);
                }
                if (!rayDone && walkOver) {
                    result = result + contrib;   // nothing in the way: transmittance 1
                    rayDone = true;
                }
                if (rayDone) {
                    r++;
                    if (!setupRay())
                        finishSlot();
                }
            }
        } else {
            bool go = busy;
            if (busy && st.leaf_batch != 1u) {           // phase vote, as in k_trace_closest_wide
                const bool wantsRecord = w.triMask != 0u;
                const uint32_t nRec = (uint32_t)__popcll(__ballot(wantsRecord)), nNode = (uint32_t)__popcll(__ballot(!wantsRecord));
                go = (nRec*st.leaf_batch >= nNode*2u) == wantsRecord;
            }
            if (go) {
                uint32_t idx = 0;
                const int what = wideNext<INST>(w, wr.octInv, stack, stride, idx);
                bool rayDone = false;
                if (what == 0) {
                    result = result + contrib;       // nothing in the way: transmittance 1
                    rayDone = true;
                } else if (INST && what == 3) {
                    const float4 sd = r == 0 ? slotF4(st, A_SH_D0, slot) : slotF4(st, A_SH_D1, slot);
                    ray.o = so; ray.d = xyz(sd); ray.tmin = eps; ray.tmax = sd.w;   // back to world space (and to the ray's own [nearT, farT])
                    wr = wideRaySetup(ray);
                    w.curInst = -1;
                } else {
                    const uint32_t off = what != 1 ? wideNodeOff(s, idx) : s.recs_offset + idx*48u;
                    const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.wide) + (size_t)off);
                    float4 q0 = p[0], q1 = p[1], q2 = p[2];
                    if (what == 2) {
                        WideNodeRegs nd;
                        wideNodeFetchRest(nd, reinterpret_cast<const char *>(s.wide), off, wr, q0, q1, q2);
                        if (COUNT) nodes++;
                        // (INST: no far distance for the nodes -- what the reference clips against the ray's farT is the box of an instance's
                        // leaf in ITS tree, tested at the record below; the geometry behind it, and so these boxes, may begin beyond farT)
                        wideVisit(w, nd, ray.o, wr, ray.tmin, INST ? PT_INF : ray.tmax);
                    } else if (INST && what == 4) {
                        wideResumeRecords(w, idx, q1);
                    } else {
                        if (COUNT) prims++;
                        if (INST && TGHIP_REC_KIND(__float_as_uint(q0.w)) == TGHIP_REC_INSTANCE) {
                            // Instance::intersect lets the ray into an instance when it passes the box of the instance's LEAF in the
                            // reference's own tree -- its test, its arithmetic (pt_kernels.h: refChildTest) -- and hands it on with nearT =
                            // the entry distance and farT = INFINITY: anything the master holds beyond occludes (instanceSetOccluded)
                            const uint32_t leaf = __float_as_uint(q2.y);
                            const float4 blo = s.inst_leaf_boxes[2u*leaf], bhi = s.inst_leaf_boxes[2u*leaf + 1u];
                            float tEntry;
                            if (refChildTest(xyz(blo), xyz(bhi), ray.o, ray.d, mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z), ray.tmin, ray.tmax, tEntry)) {
                                wideEnterInstance(w, stack, stride, idx, q0, q1, q2, ray, wr);
                                ray.tmin = tEntry; ray.tmax = PT_INF;
                            }
                        } else {
                            float tmax = ray.tmax;
                            float4 hit;
                            uint32_t meta;
                            // geometry reached through an instance belongs to the `instances` primitive, never the light (traverseOccludedInst)
                            if (testRecordLoaded<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, idx, q0, q1, q2, ray, tmax, hit, meta) && ((INST && w.curInst >= 0) || (int)TGHIP_REC_OBJECT(meta) != endCap))
                                rayDone = true;      // occluded
                        }
                    }
                }
                if (rayDone) {
                    r++;
                    if (!setupRay())
                        finishSlot();
                }
            }
            if (JOIN) PT_TURN_JOIN();
            }
    }
    if (COUNT) wpLoopEnd = wall_clock64();
    waveAddStat(&L.shadow_rays, rays);
    waveAddStat(&L.shadow_slots, slots);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }

    queuesEnd(L, st, Q_SHADOW, appendMask);
    if (threadIdx.x == 0) {
        ctl.shadow_rays += L.shadow_rays; ctl.shadow_slots += L.shadow_slots;
        if (COUNT) {
            BlockStats &bs = st.stats[blockIdx.x];
            bs.nodes_visited += L.nodes; bs.prims_tested += L.prims;
            bs.nodes_visited_shadow += L.nodes; bs.prims_tested_shadow += L.prims;
        }
    }
    if (COUNT) WALK_PROF_FLUSH(1, turns - dryTurns, dryTurns);
}

// k_trace_shadow_wide for single-level scenes, rebuilt around what its waves were waiting for (count_traversal: 5.0 us per loop turn
// against 3.4 us in the closest-hit kernel for the same instruction mix): a slot's second ray used to be set up, and a finished slot's
// NEE term added to its path, by the one or two lanes that had just got there -- global loads in divergent code, whose latency the
// whole wave sat out in nearly every turn.  Here
//   * both rays of a slot are fetched when the slot is (five independent loads in the refill block): the second ray's set-up is
//     register moves;
//   * a finished slot only marks its lane; the NEE terms of all marked lanes are added in one go right before the next refill (or,
//     once the queue is dry, in the turn they finish), so their loads fly together and once per refill instead of once per turn;
//   * the walk is the DECOUPLED one of k_trace_closest_wide: a pending record AND the next node per turn.
// Same queues, same suspended-walk protocol (Q_HOLD), same results as k_trace_shadow_wide.
template<bool COUNT, bool SOLIDS, int NTS = PT_NT_TRAV, bool INST = false>
PT_DEV void traceShadowFastBody(const DeviceScene &s, const PathState &st, const PassParams &pp, BlockLds &L, uint32_t &fetchNext, int *ldsDyn)
{
    unsigned short *order = reinterpret_cast<unsigned short *>(ldsDyn);
    uint2 *stack = reinterpret_cast<uint2 *>(ldsDyn + (st.slots_per_block >> 1)) + threadIdx.x;
    const int stride = (int)blockDim.x;
    BlockCtl &ctl = st.ctl[blockIdx.x];
    if (threadIdx.x == 0) fetchNext = 0;
    const unsigned long long wpStart = COUNT ? wall_clock64() : 0ull;
    uint32_t turns = 0, dryTurns = 0;
    const uint32_t appendMask = (1u << Q_FIN) | (!INST && st.suspend_lanes != 0u ? (1u << Q_EXT) | (1u << Q_HOLD) : 0u);
    // the top of the tree in LDS, behind the stacks (PathState::lds_nodes; queuesBegin's barriers publish the copy)
    const uint32_t topCount = st.lds_nodes;
    const char *ldsTop = reinterpret_cast<const char *>(ldsDyn) + st.slots_per_block*2u + st.wide_depth*blockDim.x*8u;
    {
        float4 *dst = reinterpret_cast<float4 *>(const_cast<char *>(ldsTop));
        for (uint32_t i = threadIdx.x; i < topCount*s.wide_stride/16u; i += blockDim.x)
            dst[i] = s.wide[i];
    }
    queuesBegin(L, st, ctl, Q_SHADOW, appendMask, order);
    const uint32_t n = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const int minBounces = s.settings.min_bounces;
    uint32_t nodes = 0, prims = 0, rays = 0, slots = 0;

    bool busy = false, pendingFinish = false, held = false;
    uint32_t slot = 0, local = 0;
    int r = 0;                                   // ray of the slot being traced (0: light sample, 1: bsdf sample)
    f3 so = splat3(0.0f);
    float eps = 0.0f;
    f3 result = splat3(0.0f);
    float4 c1 = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)), d1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // the slot's second ray
    RayD ray; ray.o = splat3(0.0f); ray.d = splat3(1.0f); ray.tmin = 0.0f; ray.tmax = 0.0f;
    WideRay wr; wr.idir = splat3(1.0f); wr.octInv = 0u;
    WideState w;
    wideStart(w);
    f3 contrib = splat3(0.0f);
    int endCap = -1;
    bool exhausted = false;
    uint32_t age = 0;
    const bool maySuspend = !INST && st.suspend_lanes != 0u && n >= st.suspend_min_queue;   // (two-level walks are not suspended)
    f3 wdir = splat3(1.0f);                      // INST: the ray being traced in world space (the walk's `ray` is in a master's space inside an instance)
    float wtmax = 0.0f;

    WALK_PROF_DECL;
    // the lane takes ray (c, sd) if it has to be traced (k_trace_shadow_wide: setupRay)
    auto tryRay = [&](float4 c, float4 sd, bool resume) -> bool {
        const uint32_t tag = __float_as_uint(c.w);
        if (tag == 0xFFFFFFFFu && !resume)
            return false;
        endCap = (int)(tag & 0xFFFFFFu);
        if (!resume) {
            rays++;
            if ((int)(tag >> 24) < minBounces)
                return false;                    // contributes nothing (TraceBase.cpp:114-115 with minBounces)
        }
        contrib = xyz(c);
        ray.o = so; ray.d = xyz(sd); ray.tmin = eps; ray.tmax = sd.w;
        if (INST) { wdir = ray.d; wtmax = ray.tmax; }
        if (!resume && s.hoisted_rec >= 0) {     // the scene's one quad, before the walk (DeviceScene::hoisted_rec): occluded by it, the ray adds nothing
            float tq = ray.tmax;
            float4 hq;
            uint32_t meta;
            if (COUNT) prims++;
            if (testRecord<true, KIND_BIT(TGHIP_REC_QUAD)>(s, (uint32_t)s.hoisted_rec, ray, tq, hq, meta) && (int)TGHIP_REC_OBJECT(meta) != endCap)
                return false;
        }
        if (COUNT && !resume) wpRays++;
        wr = wideRaySetup(ray);
        if (!resume) wideStart(w);
        return true;
    };
#if PT_SHADOW_PREP
    // Round 6: the slot's SECOND ray is prepared when the slot is fetched, not when its first ray ends.  "On to the second ray" used to run tryRay --
    // the quad test with its division, the three divisions of the reciprocal direction, ~110 instructions -- inside the loop for the two to five
    // lanes whose first ray had just ended, in nearly every turn of the wave; with eight or fewer lanes enabled a VALU instruction issues at a quarter
    // of the rate (profiles/r6_ubench_lane_masks.txt), so those lanes cost the wave about as much as the rest of its turn.  In the refill block
    // some 28 lanes are enabled (roofline.valu.walk.shadow.refill), and the step to the second ray is a handful of selects every lane executes.
    // What is prepared: whether the ray is traced at all (tag, minBounces, the hoisted quad), its reciprocal direction and octant.
    bool valid1 = false;                         // the second ray has to be traced
    f3 idir1 = splat3(1.0f);
    uint32_t oct1 = 0u;
    auto prepareSecond = [&](bool count) {
        const uint32_t tag = __float_as_uint(c1.w);
        valid1 = tag != 0xFFFFFFFFu;
        if (valid1 && count) rays++;
        valid1 = valid1 && (int)(tag >> 24) >= minBounces;
        RayD r1; r1.o = so; r1.d = xyz(d1); r1.tmin = eps; r1.tmax = d1.w;
        if (s.hoisted_rec >= 0) {
            float tq = r1.tmax;
            float4 hq;
            uint32_t meta;
            if (COUNT && valid1 && count) prims++;
            if (testRecord<true, KIND_BIT(TGHIP_REC_QUAD)>(s, (uint32_t)s.hoisted_rec, r1, tq, hq, meta) && (int)TGHIP_REC_OBJECT(meta) != (int)(tag & 0xFFFFFFu))
                valid1 = false;
        }
        if (COUNT && valid1 && count) wpRays++;
        const WideRay w1 = wideRaySetup(r1);
        idir1 = w1.idir; oct1 = w1.octInv;
    };
    // ray r is done: on to the slot's second ray, or the slot is finished (its NEE term is added at the next refill) -- selects, for every lane
    auto nextRay = [&](bool done) {
        const bool second = done && r == 0 && valid1;
        r = done ? 1 : r;
        endCap = second ? (int)(__float_as_uint(c1.w) & 0xFFFFFFu) : endCap;
        contrib = sel3(second, xyz(c1), contrib);
        ray.o = sel3(second, so, ray.o);         // (a slot whose FIRST ray is not traced never went through tryRay's assignments)
        ray.tmin = second ? eps : ray.tmin;
        ray.d = sel3(second, xyz(d1), ray.d);
        ray.tmax = second ? d1.w : ray.tmax;
        wr.idir = sel3(second, idir1, wr.idir);
        wr.octInv = second ? oct1 : wr.octInv;
        wideStartIf(w, second);
        if (INST) { wdir = sel3(second, xyz(d1), wdir); wtmax = second ? d1.w : wtmax; w.curInst = second ? -1 : w.curInst; w.curNode = second ? 0u : w.curNode; }
        busy = (done && !second) ? false : busy;
        pendingFinish = (done && !second) ? true : pendingFinish;
    };
#else
    // ray r is done: on to the slot's second ray, or the slot is finished (its NEE term is added at the next refill)
    auto nextRay = [&](bool done) {
        if (!done) return;
        if (r == 0) {
            r = 1;
            if (tryRay(c1, d1, false))
                return;
        }
        busy = false;
        pendingFinish = true;
    };
#endif

    for (;;) {
        unsigned long long busyMask = __ballot(busy);
        const bool refill = !exhausted && __popcll(busyMask) <= PT_REFILL_AT;
        const unsigned long long pendMask = __ballot(pendingFinish);   // (after the queue ran dry: more than eight lanes at a time, traceClosestWideBody)
        if (pendMask != 0ull && (refill || (exhausted && (!PT_LATE_PUBLISH || __popcll(pendMask) > 8 || busyMask == 0ull)))) {
            WALK_SECTION(wpPubs, wpPubLanes, pendingFinish);
            if (pendingFinish) {
                // NEE term -> path radiance; paths that ended at this vertex go on the finished list
                const float4 wgt = slotF4<NTS>(st, A_SH_W, slot), p = slotF4<NTS>(st, A_SH_P, slot), e4 = slotF4<NTS>(st, A_EMI, slot);
                f3 em = xyz(e4);
                em = em + (result*wgt.w)*xyz(wgt);           // emission += estimateDirect(...)*throughput
                em = em + xyz(p);
                slotF4<NTS>(st, A_EMI, slot) = mk4(em, 0.0f);
                queuePush(FLAG_STATE(__float_as_uint(p.w)) != ST_ACTIVE, local, L, Q_FIN);
                if (held) {                                  // the path's extension ray was held back while this slot was suspended: release it
                    const uint32_t bit = 1u << (local & 31u);
                    if (L.bm[Q_HOLD][local >> 5] & bit) {
                        atomicAnd(&L.bm[Q_HOLD][local >> 5], ~bit);
                        atomicOr(&L.bm[Q_EXT][local >> 5], bit);
                    }
                }
                pendingFinish = false;
            }
        }
        if (refill) {
            unsigned long long want = ~busyMask;
            uint32_t lane = laneId();
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&fetchNext, (uint32_t)__popcll(want));
            base = __shfl(base, 0);
            WALK_SECTION(wpRefills, wpRefillLanes, !busy && base + __popcll(want & ((1ull << lane) - 1ull)) < n);
            if (!busy) {
                uint32_t i = base + __popcll(want & ((1ull << lane) - 1ull));
                if (i < n) {
                    local = order[i];
                    slot = first + local;
                    const float4 o4 = slotF4<NTS>(st, A_SH_O, slot), c0 = slotF4<NTS>(st, A_SH_C0, slot), d0 = slotF4<NTS>(st, A_SH_D0, slot);
                    c1 = slotF4<NTS>(st, A_SH_C1, slot); d1 = slotF4<NTS>(st, A_SH_D1, slot);
                    so = xyz(o4);
                    const bool resumed = (__float_as_uint(o4.w) & WALK_SUSPENDED_BIT) != 0u;   // a slot an earlier launch suspended (below)
                    eps = __uint_as_float(__float_as_uint(o4.w) & ~WALK_SUSPENDED_BIT);
                    held = resumed;
                    age = 0;
                    busy = true;
                    if (!resumed) {
                        slots++;
                        result = splat3(0.0f);
                        r = 0;
#if PT_SHADOW_PREP
                        prepareSecond(true);
                        nextRay(!tryRay(c0, d0, false));
#else
                        if (!tryRay(c0, d0, false))
                            nextRay(true);
#endif
                    } else {
                        walkRestore(st, slot, w, stack, stride);
                        if (COUNT) wpResumed++;
                        const float4 part = slotF4<NTS>(st, st.walk_base + 2u, slot);
                        result = xyz(part); r = __float_as_int(part.w);
                        slotW(st, A_SH_O, slot, 3u) = eps;                   // (an ordinary shadow slot again)
                        (void)tryRay(r == 0 ? c0 : c1, r == 0 ? d0 : d1, true);
#if PT_SHADOW_PREP
                        prepareSecond(false);    // (a resumed first ray: its slot's second ray was counted when the slot was first fetched)
#endif
                    }
                }
            }
            if (base + (uint32_t)__popcll(want) >= n) {
                exhausted = true;
                if (COUNT) wpDry = wall_clock64();
            }
            busyMask = __ballot(busy);
        }
        // the queue is dry and the wave is down to its last, longest walks: suspend those that have had their turns (k_trace_closest_wide).
        // The slot stays on the shadow queue; its path's extension ray, if it has one, waits in Q_HOLD until the slot is resolved.
        if (maySuspend && exhausted && (uint32_t)__popcll(busyMask) <= st.suspend_lanes) {
            if (busy && age >= st.suspend_turns) {
                walkSave(st, slot, w, stack, stride);
                if (COUNT) wpSuspended++;
                slotF4<NTS>(st, st.walk_base + 2u, slot) = mk4(result, __int_as_float(r));
                slotW(st, A_SH_O, slot, 3u) = __uint_as_float(__float_as_uint(eps) | WALK_SUSPENDED_BIT);
                queuePush(true, local, L, Q_SHADOW);
                const uint32_t bit = 1u << (local & 31u);
                if (L.bm[Q_EXT][local >> 5] & bit) {
                    atomicAnd(&L.bm[Q_EXT][local >> 5], ~bit);
                    atomicOr(&L.bm[Q_HOLD][local >> 5], bit);
                }
                busy = false;
            }
            busyMask = __ballot(busy);
        }
        if (busyMask == 0ull) {
            if (__ballot(pendingFinish) != 0ull)
                continue;                        // (the slots that finished in the last turn: added at the top of the loop)
            break;
        }
        age++;
        if (COUNT) { turns++; dryTurns += exhausted ? 1u : 0u; if (exhausted) wpBusyDry += (uint32_t)__popcll(busyMask); else wpBusy += (uint32_t)__popcll(busyMask); }
        if constexpr (INST) {
            // Two-level scenes: the walk of k_trace_shadow_wide<., ., INST> -- one node or one record per lane and turn, a vote on which of the two a
            // turn runs, instance records entered by the reference's leaf test -- inside this kernel's slot handling: both rays of a slot fetched at
            // the refill, the second one prepared there, finished slots added to their paths in one go (round 6; that kernel set up a slot's next ray
            // and finished slots with global loads in divergent code, for the one to five lanes that had just got there, in nearly every turn: 0.19
            // of the lanes of an issued VALU instruction enabled, profiles/r6_sq_counters_instances10k.json).
            bool go = busy;
            if (busy && st.leaf_batch != 1u) {           // phase vote, as in k_trace_closest_wide
                const bool wantsRecord = w.triMask != 0u;
                const uint32_t nRec = (uint32_t)__popcll(__ballot(wantsRecord)), nNode = (uint32_t)__popcll(__ballot(!wantsRecord));
                go = (nRec*st.leaf_batch >= nNode*2u) == wantsRecord;
            }
            bool rayDone = false;
            if (go) {
                uint32_t idx = 0;
                const int what = wideNext<true>(w, wr.octInv, stack, stride, idx);
                if (what == 0) {
                    result = result + contrib;       // nothing in the way: transmittance 1
                    rayDone = true;
                } else if (what == 3) {
                    ray.o = so; ray.d = wdir; ray.tmin = eps; ray.tmax = wtmax;     // back to world space (and to the ray's own [nearT, farT])
                    wr = wideRaySetup(ray);
                    w.curInst = -1;
                } else {
                    const uint32_t off = what != 1 ? wideNodeOff(s, idx) : s.recs_offset + idx*48u;
                    const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.wide) + (size_t)off);
                    float4 q0 = p[0], q1 = p[1], q2 = p[2];
                    if (what == 2) {
                        WideNodeRegs nd;
                        wideNodeFetchRest(nd, reinterpret_cast<const char *>(s.wide), off, wr, q0, q1, q2);
                        if (COUNT) nodes++;
                        // (no far distance for the nodes -- what the reference clips against the ray's farT is the box of an instance's leaf in ITS
                        // tree, tested at the record below; the geometry behind it, and so these boxes, may begin beyond farT)
                        wideVisit(w, nd, ray.o, wr, ray.tmin, PT_INF);
                    } else if (what == 4) {
                        wideResumeRecords(w, idx, q1);
                    } else {
                        if (COUNT) prims++;
                        if (TGHIP_REC_KIND(__float_as_uint(q0.w)) == TGHIP_REC_INSTANCE) {
                            // Instance::intersect lets the ray into an instance when it passes the box of the instance's LEAF in the reference's own
                            // tree -- its test, its arithmetic (pt_kernels.h: refChildTest) -- and hands it on with nearT = the entry distance and
                            // farT = INFINITY: anything the master holds beyond occludes (instanceSetOccluded)
                            const uint32_t leaf = __float_as_uint(q2.y);
                            const float4 blo = s.inst_leaf_boxes[2u*leaf], bhi = s.inst_leaf_boxes[2u*leaf + 1u];
                            float tEntry;
                            if (refChildTest(xyz(blo), xyz(bhi), ray.o, ray.d, mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z), ray.tmin, ray.tmax, tEntry)) {
                                wideEnterInstance(w, stack, stride, idx, q0, q1, q2, ray, wr);
                                ray.tmin = tEntry; ray.tmax = PT_INF;
                            }
                        } else {
                            float tmax = ray.tmax;
                            float4 hit;
                            uint32_t meta;
                            // geometry reached through an instance belongs to the `instances` primitive, never the light (traverseOccludedInst)
                            if (testRecordLoaded<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, idx, q0, q1, q2, ray, tmax, hit, meta) && (w.curInst >= 0 || (int)TGHIP_REC_OBJECT(meta) != endCap))
                                rayDone = true;      // occluded
                        }
                    }
                }
            }
            nextRay(busy && rayDone);
            PT_TURN_JOIN();
        } else
        if (busy) {
            uint32_t recIdx = 0, nodeIdx = 0;
            bool hasRec = false, hasNode = false;
            if (w.triMask == 0u && w.tri2Mask != 0u) { w.triBase = w.tri2Base; w.triMask = w.tri2Mask; w.triValid = w.tri2Valid; w.tri2Mask = 0u; }
            if (w.triMask) {
                const uint32_t b = (uint32_t)__ffs((int)w.triMask) - 1u;
                recIdx = w.triBase + (uint32_t)__popc(w.triValid & ((1u << b) - 1u));
                w.triMask &= w.triMask - 1u;
                hasRec = true;
            }
            if (w.tri2Mask == 0u)
                hasNode = wideNextNode(w, wr.octInv, stack, stride, nodeIdx);
            float4 r0, r1, r2;
            WideNodeRegs nd;
            PT_WALK_FETCH(s, st, r0, r1, r2, nd, hasRec, recIdx, hasNode, nodeIdx, wr, topCount, ldsTop);
            WALK_SECTION(wpRecTurns, wpRecLanes, hasRec);       // (inside `if (busy)`: the ballot covers the lanes that are here)
            WALK_SECTION(wpNodeTurns, wpNodeLanes, hasNode);
            bool rayDone = false;
            if (hasRec) {
                if (COUNT) prims++;
                float tmax = ray.tmax;
                float4 hit;
                uint32_t meta;
                const bool accepted = testRecordLoaded<false, SOLIDS ? KINDS_ALL : KINDS_MESH>(s, recIdx, r0, r1, r2, ray, tmax, hit, meta);
                if (COUNT && accepted) wpHits++;
                if (accepted && (int)TGHIP_REC_OBJECT(meta) != endCap)
                    rayDone = true;              // occluded
            }
            if (!rayDone && hasNode) {
                if (COUNT) nodes++;
                const uint32_t ob = w.triBase, om = w.triMask, ov = w.triValid;
                wideVisit(w, nd, ray.o, wr, ray.tmin, ray.tmax, s.hoisted_rec >= 0);
                if (om) { w.tri2Base = w.triBase; w.tri2Mask = w.triMask; w.tri2Valid = w.triValid; w.triBase = ob; w.triMask = om; w.triValid = ov; }
            }
            if (!rayDone && wideWalkOver(w)) {
                result = result + contrib;       // nothing in the way: transmittance 1
                rayDone = true;
            }
            nextRay(rayDone);
        }
    }
    if (COUNT) wpLoopEnd = wall_clock64();
    waveAddStat(&L.shadow_rays, rays);
    waveAddStat(&L.shadow_slots, slots);
    if (COUNT) { waveAddStat(&L.nodes, nodes); waveAddStat(&L.prims, prims); }

    queuesEnd(L, st, Q_SHADOW, appendMask);
    if (threadIdx.x == 0) {
        ctl.shadow_rays += L.shadow_rays; ctl.shadow_slots += L.shadow_slots;
        if (COUNT) {
            BlockStats &bs = st.stats[blockIdx.x];
            bs.nodes_visited += L.nodes; bs.prims_tested += L.prims;
            bs.nodes_visited_shadow += L.nodes; bs.prims_tested_shadow += L.prims;
        }
    }
    if (COUNT) WALK_PROF_FLUSH(1, turns - dryTurns, dryTurns);
}
template<bool COUNT, bool SOLIDS = true>
__global__ WIDE_SHADOW_BOUNDS void k_trace_shadow_fast(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    traceShadowFastBody<COUNT, SOLIDS>(s, st, pp, L, fetchNext, ldsDyn);
}
// ... and for scenes with instance records (round 6): the two-level wide walk inside the same slot handling
template<bool COUNT, bool SOLIDS = true>
__global__ WIDE_SHADOW_BOUNDS void k_trace_shadow_fast_inst(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    traceShadowFastBody<COUNT, SOLIDS, PT_NT_OTHER, true>(s, st, pp, L, fetchNext, ldsDyn);
}

// Second half of the dynamic-fetch shadow step: finalises the paths that had ended at the vertex whose shadow rays
// k_trace_shadow_dyn just resolved (Q_FIN), regenerates their slots and reports whether the workgroup has extension
// rays for the next iteration (returned; BlockCtl::live_slots = how many of its slots still carry a path).
template<int NTS, bool EXT>
PT_DEV bool finishBody(const DeviceScene &s, const PathState &st, const PassParams &pp, BlockLds &L, unsigned short *order)
{
    BlockCtl &ctl = st.ctl[blockIdx.x];
    // (Q_SHADOW is loaded for the liveness report only: it holds the slots k_trace_shadow_wide suspended, whose paths -- waiting in
    // Q_HOLD or already ended -- keep the pass alive although no extension ray may be queued)
    queuesBegin(L, st, ctl, Q_FIN, (1u << Q_EXT) | (1u << Q_EXTP) | (1u << Q_SHADOW), order);
    const uint32_t nf = L.n;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    const bool aborted = __hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
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
            em = xyz(slotF4<NTS>(st, A_EMI, sl));
            black = FLAG_STATE(__float_as_uint(slotW(st, A_SH_P, sl, 3u))) == ST_TERMINATED_BLACK;
        }
        bool regenerated = nextPath<true, EXT, NTS>(s, st, pp, fin, false, sl, em, black, &L.cursor, aborted, finishedCount);
        queuePush(regenerated, loc, L, Q_EXTP);
    }
    waveAddStat(&L.samples, finishedCount);
    const bool anyExt = queuesEnd(L, st, Q_FIN, (1u << Q_EXT) | (1u << Q_EXTP) | (1u << Q_SHADOW), -1, false, 1u << Q_SHADOW);
    // paths the workgroup still carries: the extension rays queued for the next iteration and the suspended shadow slots
    uint32_t liveSlots = 0;
    for (uint32_t wd = threadIdx.x; wd < (st.slots_per_block >> 5); wd += blockDim.x)
        liveSlots += (uint32_t)__popc(L.bm[Q_EXT][wd] | L.bm[Q_EXTP][wd] | L.bm[Q_SHADOW][wd]);
    waveAddStat(&L.nodes, liveSlots);            // (L.nodes: unused by this kernel otherwise)
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl.item_cursor = L.cursor;
        ctl.samples += L.samples;
        ctl.live_slots = L.nodes;
    }
    return anyExt;
}
#ifdef PT_WAVEFRONT_MAIN   /* non-template kernels live in the shim's translation unit only */
__global__ __launch_bounds__(256) void k_finish(DeviceScene s, PathState st, PassParams pp, uint32_t iterTag)
{
    __shared__ BlockLds L;
    __shared__ unsigned short order[PT_MAX_SLOTS_PER_BLOCK];
    const bool anyExt = finishBody(s, st, pp, L, order);
    if (threadIdx.x == 0 && anyExt)
        atomicMax(&st.live[0], iterTag);         // (max: the parts of the pool run on streams of their own and pass here out of order)
}
// sums BlockCtl::live_slots over the workgroups of the pool into live[1] (launched with one workgroup at every host check)
__global__ __launch_bounds__(256) void k_live_slots(PathState st, uint32_t grid)
{
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    uint32_t sum = 0;
    for (uint32_t b = threadIdx.x; b < grid; b += blockDim.x)
        sum += st.ctl[b].live_slots;
    waveAddStat(&total, sum);
    __syncthreads();
    if (threadIdx.x == 0) st.live[1] = total;
}
#endif

// The tail of a pass in ONE launch per part of the pool: every workgroup runs the wavefront iterations of its own slots -- closest hit,
// the shading classes, shadow rays, finish / regenerate: the bodies of the kernels above, one after the other, through the same queue
// bitmaps -- until its queues are empty.  Nothing a workgroup reads or writes in an iteration is shared with another workgroup, so the
// iterations need no grid-wide synchronisation; what the launch saves is the latency of 7 near-empty launches per iteration for the ~30
// iterations in which the last, longest paths of a pass (up to 64 bounces) run out -- the shim switches to it when few paths are left
// (tghip_ctx::tailThreshold).  One shading variant (M: every BSDF type of the scene's classes) shades all classes; results are those of
// the per-class launches bit for bit (tests/test_gpu_parity.py).
template<uint32_t M, bool SOLIDS>
__global__ __launch_bounds__(256) void k_tail(DeviceScene s, PathState st, PassParams pp, uint32_t classes)
{
    extern __shared__ int ldsDyn[];
    __shared__ BlockLds L;
    __shared__ uint32_t fetchNext;
    __shared__ __attribute__((aligned(16))) unsigned char ldsTables[PT_LDS_TABLE_BYTES];
    __shared__ unsigned short order[PT_MAX_SLOTS_PER_BLOCK];
    const DeviceScene staged = stageSceneTables(s, ldsTables);   // (for the shading steps; the traversal steps read the scene's big arrays only)
    constexpr int TNT = PT_NT_TAIL ? PT_NT_STATE : 0;            // (a tail workgroup re-reads what it wrote a few microseconds ago: plain stores)
    for (;;) {
        traceClosestWideBody<false, SOLIDS, false, true, TNT>(s, st, L, fetchNext, ldsDyn);
        __syncthreads();
        for (int c = 0; c < PT_NUM_CLASSES; ++c) {           // class 0 with the escaped paths, then the classes that occur in the scene (bit c of `classes`)
            if (c >= 1 && !((classes >> c) & 1u))
                continue;
            (void)shadeBody<M, 0, true>(staged, st, pp, c == 0 ? CLS_0_AND_MISS : c, L, ldsTables, order);
            __syncthreads();
        }
        traceShadowFastBody<false, SOLIDS, TNT>(s, st, pp, L, fetchNext, ldsDyn);
        __syncthreads();
        const bool anyExt = finishBody<TNT>(s, st, pp, L, order);
        __syncthreads();
        if (!anyExt)
            break;
    }
}

// Sums the per-item partial sums of every pixel slot in fixed chunk order into the framebuffer
// (deterministic; no float atomics anywhere on the accumulation path).
#ifdef PT_WAVEFRONT_MAIN   /* non-template kernels live in the shim's translation unit only */
__global__ __launch_bounds__(256) void k_resolve(PathState st, PassParams pp, float *fbSum, uint32_t *fbCount)
{
    uint32_t j = blockIdx.x*blockDim.x + threadIdx.x;
    if (j >= pp.pix_slots)
        return;
    uint32_t x, y;
    if (!slotPixel(pp, j, x, y))
        return;
    uint32_t pixel = x + y*pp.width;
    // One order of additions for a pixel's samples, whatever the items, batches and shards they were traced in: groups of four consecutive
    // samples (counted from the pass's first) are summed first to last -- inside a four-sample item's slot, or here over four one-sample
    // items --, and the groups are added to the pixel one after the other.  Batches hold whole groups, so splitting a pass changes nothing.
    float sx = fbSum[(size_t)pixel*3 + 0], sy = fbSum[(size_t)pixel*3 + 1], sz = fbSum[(size_t)pixel*3 + 2];
    uint32_t cnt = 0;
    const uint32_t group = pp.group ? pp.group : 1u;
    for (uint32_t c = 0; c < pp.chunks; c += group) {
        float4 g = at32(st.partial, c*pp.pix_slots + j);
        cnt += __float_as_uint(g.w);
        for (uint32_t k = 1; k < group && c + k < pp.chunks; ++k) {
            float4 a = at32(st.partial, (c + k)*pp.pix_slots + j);
            g.x += a.x; g.y += a.y; g.z += a.z;
            cnt += __float_as_uint(a.w);
        }
        sx += g.x; sy += g.y; sz += g.z;
    }
    fbSum[(size_t)pixel*3 + 0] = sx;
    fbSum[(size_t)pixel*3 + 1] = sy;
    fbSum[(size_t)pixel*3 + 2] = sz;
    fbCount[pixel] += cnt;
}
#endif

// k_resolve of a record pass (gap-free item enumeration, PassParams): one thread per pixel slot of the sorted record list
// sums the pixel's items of this batch in chunk order.
#ifdef PT_WAVEFRONT_MAIN   /* non-template kernels live in the shim's translation unit only */
__global__ __launch_bounds__(256) void k_resolve_records(PathState st, PassParams pp, float *fbSum, uint32_t *fbCount)
{
    uint32_t j = blockIdx.x*blockDim.x + threadIdx.x;
    if (j >= pp.num_sorted*16u)
        return;
    uint32_t rec = pp.rec_sorted[j >> 4];
    uint32_t x = (rec % pp.variance_w)*4u + (j & 3u), y = (rec/pp.variance_w)*4u + ((j >> 2) & 3u);
    if (x >= pp.width || y >= pp.height)
        return;
    // (the order of additions of k_resolve: groups of pp.group items first, group after group into the pixel; a batch holds whole groups)
    uint32_t pixel = x + y*pp.width;
    float sx = fbSum[(size_t)pixel*3 + 0], sy = fbSum[(size_t)pixel*3 + 1], sz = fbSum[(size_t)pixel*3 + 2];
    uint32_t cnt = 0;
    bool done = false;
    const uint32_t group = pp.group ? pp.group : 1u;
    for (uint32_t c0 = 0; c0 < pp.num_chunks && !done; c0 += group) {
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
        bool any = false;
        for (uint32_t c = c0; c < c0 + group && c < pp.num_chunks; ++c) {
            uint32_t start = pp.rec_chunk_start[c];
            if (j >= pp.rec_chunk_start[c + 1u] - start) { done = true; break; }   // chunks only get shorter: the pixel has no further items
            uint32_t w = start + j;
            if (w < pp.item_base || w - pp.item_base >= pp.total_items)
                continue;                        // item of another batch
            float4 a = at32(st.partial, w - pp.item_base);
            if (!any) { gx = a.x; gy = a.y; gz = a.z; any = true; }
            else { gx += a.x; gy += a.y; gz += a.z; }
            cnt += __float_as_uint(a.w);
        }
        if (any) { sx += gx; sy += gy; sz += gz; }
    }
    fbSum[(size_t)pixel*3 + 0] = sx;
    fbSum[(size_t)pixel*3 + 1] = sy;
    fbSum[(size_t)pixel*3 + 2] = sz;
    fbCount[pixel] += cnt;
}
#endif

// SampleRecord::addSample (path_tracer/SampleRecord.hpp:46-58) over the luminances the pass wrote, one thread per
// record, in the order the reference's renderTile visits them (PathTraceIntegrator.cpp:136-156: the tile's pixels
// row by row, each pixel's samples in index order), so mean and running variance round exactly like the CPU's.
#ifdef PT_WAVEFRONT_MAIN   /* non-template kernels live in the shim's translation unit only */
__global__ __launch_bounds__(64) void k_records(PassParams pp, TgHipSampleRecord *records, uint32_t numRecords)
{
    uint32_t r = blockIdx.x*blockDim.x + threadIdx.x;
    if (r >= numRecords)
        return;
    uint32_t rx = r % pp.variance_w, ry = r/pp.variance_w;
    if (pp.shard_count > 1u && ((rx >> 2) + (ry >> 2)*pp.shard_skew) % pp.shard_count != pp.shard_index)   // tghip_tile_owner
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
#endif

#endif
