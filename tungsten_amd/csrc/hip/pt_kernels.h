// Wavefront path-tracing kernels for gfx950 (MI355X).  Replaces, per (pixel, sample):
//   PathTraceIntegrator::renderTile            (integrators/path_tracer/PathTraceIntegrator.cpp:136-156)
//   PathTracer::traceSample                    (integrators/path_tracer/PathTracer.cpp:14-149)
//   TraceBase::{handleSurface, estimateDirect, lightSample, bsdfSample, generalizedShadowRay,
//               handleInfiniteLights}          (integrators/TraceBase.cpp)
//   TraceableScene::intersect + Embree         (renderer/TraceableScene.hpp:170-192)
//
// Execution model: the machine is partitioned into G persistent workgroups (G = CUs x
// blocks_per_cu, the same G for every kernel of a pass).  Workgroup b owns
//   * a private range of path slots in HBM (slot = b*slots_per_block + local; 17 arrays of 16 B per slot in one
//     allocation, addressed as "one base + 32-bit offset"),
//   * private queues over those slots -- one BITMAP per queue, expanded by the consumer into the ascending list of
//     set slots so that every kernel walks its slots in memory order -- and
//   * a private, finely interleaved share of the pass's work items (item = pixel x chunk of sample indices).
// Nothing is shared between workgroups, so there is not a single global atomic on the hot path: queue pushes are
// LDS bit-sets, work-item fetches wave-aggregated LDS atomics, and queue contents travel from one kernel to the next
// as per-workgroup bitmaps.  (Same-address global atomics saturate at ~88/us on MI355X -- MI355X_MICROARCH.md
// "dequeue" -- which is what bounded the first version of this tracer.)
// A slot that finishes its item flushes the item's radiance sum to partial[item] and takes the workgroup's next
// item, so the pool stays full until the pass drains however uneven the path lengths are; partial[] is reduced
// per pixel in fixed chunk order by k_resolve, which keeps the image bit-reproducible.
// One wavefront iteration of a BVH scene is
//     k_trace_closest_dyn -> k_shade<simple> [-> k_shade<complex>] -> k_trace_shadow_dyn
//   k_trace_closest_dyn  BVH2 closest hit (per-lane node stack in LDS, dynamic ray fetch); bins each path by the
//                        shading class of the surface it hit into one queue per class ("sort by material").
//   k_shade<M, W, FUSE>  handleSurface for one class, compiled for the BSDF type / scene feature set M only; emits
//                        <= 2 shadow rays and the continuation ray; finished paths are finalised and regenerated
//                        in place.
//   k_trace_shadow_dyn   generalizedShadowRay for the queued shadow rays (any-hit, dynamic fetch); finishes the
//                        paths that were waiting for their last shadow result.
// Flat-list scenes (<= TGHIP_FLAT_MAX_RECS records) intersect inside k_shade (FUSE flags) and, without class-1
// materials, run every workgroup to completion in a single launch.
#ifndef TGAMD_PT_KERNELS_H_
#define TGAMD_PT_KERNELS_H_

#include "pt_scene.h"

// ---- slot state ------------------------------------------------------------------------------
enum { ST_DONE = 0, ST_ACTIVE = 1, ST_TERMINATED = 2, ST_TERMINATED_BLACK = 3 };
#define FLAG_BOUNCE(f)      ((f) & 0xFFu)
#define FLAG_SPECULAR       0x100u
#define FLAG_STATE(f)       (((f) >> 16) & 7u)
#define FLAG_MAKE(bounce, spec, state) ((uint32_t)(bounce) | ((spec) ? FLAG_SPECULAR : 0u) | ((uint32_t)(state) << 16))
// media scenes (FEAT_MEDIA variants only): the medium the path is in, as index + 1 (0 = none), and MediumState::bounce
// (scatter events since the last surface, Medium.hpp:30-47)
#define FLAG_MEDIUM(f)         ((int)(((f) >> 9) & 0x7Fu) - 1)
#define FLAG_MEDIUM_BOUNCE(f)  (((f) >> 19) & 0xFFu)
#define FLAG_MEDIUM_BITS(medium, mbounce) ((((uint32_t)((medium) + 1)) & 0x7Fu) << 9 | (((uint32_t)(mbounce)) & 0xFFu) << 19)
#define FLAG_AUX_RECORDED      (1u << 27)   /* TGHIP_PASS_AUX: recordedOutputValues (PathTracer.cpp:46) */
#define PT_MAX_MEDIA 126u
// shadow-ray tag of a media scene: light object (16 bits) | medium the ray starts in + 1 (8 bits) | bounce (8 bits)
#define SHADOW_TAG_MEDIA(light, medium, bounce) ((uint32_t)(light) | ((uint32_t)((medium) + 1) << 16) | ((uint32_t)(bounce) << 24))

// Shading classes ("sort by material"): the class of a record's BSDF says which k_shade variant shades a hit on it -- 0: Lambert / null
// (MASK_SIMPLE), 1: the conductor family (MASK_COAT: rough conductor, conductor, mirror, smooth coat over those), 2: the dielectric
// family (MASK_GLASS: dielectric, rough dielectric), 3: everything else (plastics, mixed, transparency, forward: MASK_PLASTIC when that
// covers them, else every type); CLS_MISS: the path's ray left the scene.  One queue and one launch per class that occurs in the scene.
#define PT_NUM_CLASSES 4
#define CLS_MISS 4
#define CLS_0_AND_MISS 5   /* a k_shade launch that consumes the queue of class 0 and, behind it, the escaped paths' (same shading variant) */
#ifndef PT_ITEM_GROUP
#define PT_ITEM_GROUP  64u        // consecutive work items handed to one workgroup (a wave's worth of pixels)
#endif
#ifndef PT_MAX_SLOTS_PER_BLOCK
#define PT_MAX_SLOTS_PER_BLOCK 4096u
#endif
#define PT_MAX_WORDS   (PT_MAX_SLOTS_PER_BLOCK/32u)

// The four queues of a workgroup are BITMAPS over its slot range (one bit per slot), not index lists: a consumer
// expands its bitmap into the ascending list of set slots, so every kernel walks its slots in memory order and
// the 16-byte-per-slot SoA accesses of a wave land in consecutive cache lines.  (Index lists in arrival order
// made each wave touch ~50 different 64-B lines per access and half of all DRAM writes partial-line.)
// Q_EXT holds bounced (continuation) rays, Q_EXTP freshly generated camera rays: the traversal kernel walks them as
// two separate runs so that the coherent primaries are not interleaved lane by lane with incoherent bounces.
// Q_FIN: paths whose last vertex was waiting for a shadow result of the dynamic-fetch shadow kernel (k_finish regenerates them)
// Q_MISS: paths whose ray left the scene -- two of five vertices of an environment-lit scene -- shaded by a launch of their own
// (handleInfiniteLights, finalise, regenerate) so that they do not sit out the surface shading of the hits in their waves
// Q_HOLD: extension rays of paths whose shadow slot is a SUSPENDED walk (below): the path must not reach its next vertex -- whose NEE
// would overwrite the slot's shadow records -- before those rays are resolved, so k_trace_shadow_wide moves its bit from Q_EXT here
// when it suspends the slot and back when the resumed walk completes
enum { Q_EXT = 0, Q_SHADE0 = 1, Q_SHADE1 = 2, Q_SHADOW = 3, Q_EXTP = 4, Q_FIN = 5, Q_MISS = 6, Q_HOLD = 7, Q_SHADE2 = 8, Q_SHADE3 = 9, Q_COUNT = 10 };
#define Q_SHADE_MASK ((1u << Q_SHADE0) | (1u << Q_SHADE1) | (1u << Q_SHADE2) | (1u << Q_SHADE3) | (1u << Q_MISS))   /* what a closest-hit kernel appends to */

struct BlockCtl {                 // one per persistent workgroup; only that workgroup touches it
    uint32_t item_cursor;         // workgroup-local linear index of the next work item
    uint32_t live_slots;          // slots that carried a path after the workgroup's last k_finish (extension queues + suspended shadow slots)
    uint32_t pad[6];
    unsigned long long samples, closest_rays, shadow_rays, shadow_slots;
};                                // 64 B

#define PT_WALK_STATS 48
struct BlockStats {               // traversal statistics (count_traversal option), one per workgroup
    unsigned long long nodes_visited, prims_tested, nodes_visited_shadow, prims_tested_shadow;
    unsigned long long prof[16];  // wave-cycles per k_shade section (only in -DPT_PROFILE builds: `make PROFILE=1`, TGHIP_VERBOSE prints them)
    // count_traversal: where the waves of the wide traversal kernels spend their time ([0] closest-hit, [1] shadow), in wall_clock64 ticks
    // (10 ns) summed over waves: 0 queue expansion, 1 loop while the workgroup's queue has rays, 2 loop after it ran dry, 3 waiting for the
    // workgroup's other waves + write-back; 4 waves; 5 / 6 loop turns before / after dry; 7 / 8 busy lanes summed over those turns;
    // 9 walks suspended, 10 walks resumed, 11 longest loop of a wave (max, ticks);
    // lane utilisation of the decoupled turn's sections (round 6; tghip_get_walk_stats): 12 / 13 wave turns that ran the record test / lanes that had a
    // record in them, 14 / 15 the same for the node visit, 16 / 17 refill blocks run / lanes refilled in them, 18 / 19 publish (closest hit) or
    // NEE-term (shadow) blocks run / lanes in them, 20 record tests accepted (a hit: the division and t, u, v), 21 rays (closest) or rays traced (shadow)
    unsigned long long walk[2][PT_WALK_STATS];
#ifdef PT_PROFILE
    unsigned long long profCls[PT_NUM_CLASSES + 2][16];   // the same per shading class of the launch (CLS_MISS = escaped paths)
    unsigned long long profLanes[PT_NUM_CLASSES + 2][16]; // ticks x enabled lanes per section and class
#endif
};

// Per-slot state: A_COUNT arrays of 16-byte elements in ONE allocation, array a at byte offset a*stride (stride =
// 16 B x pool slots).  One base pointer + one 32-bit per-lane offset address every array, which keeps the kernels'
// SGPR (base pointers) and VGPR (64-bit addresses) budgets small.
enum {
    A_RAY_O = 0,   // origin.xyz, tmin
    A_RAY_D,       // dir.xyz, tmax
    A_HIT,         // t, u, v, record index (int bits; -1 = miss)
    A_THR,         // throughput.rgb, flags (uint bits)
    A_EMI,         // radiance of the sample in flight
    A_ACC,         // sum of the finished samples of the slot's current work item, count (uint bits)
    A_MISC,        // uint4: PCG state lo, hi | pixel index | work item
    A_SAMP,        // uint4: current sample index, end of the item's sample range, -, -
    A_SH_O,        // shadow origin.xyz, epsilon
    A_SH_D0, A_SH_C0,   // light-sample shadow ray: dir.xyz, tmax | unoccluded contribution, endCap|bounce bits
    A_SH_D1, A_SH_C1,   // bsdf-sample shadow ray
    A_SH_W,        // throughput at the NEE vertex, light-selection weight
    A_SH_P,        // emission picked up at the same vertex (added after the NEE term), path flags (uint bits)
    // TGHIP_PASS_AUX passes only: what the sample in flight adds to the auxiliary output buffers; NaN = not recorded
    // (OutputBuffer::addSample drops NaN values without counting them, cameras/OutputBuffer.hpp:106-107)
    A_AUX0,        // normal.xyz | depth -- until the values are recorded (FLAG_AUX_RECORDED), .w is the running hitDistance
    A_AUX1,        // albedo.rgb | visibility (+inf = waiting for the light sample's shadow ray of the recording vertex)
    // PathState::nee_factors scenes only (forward lobes, media, mesh emitters: the shadow rays' transmittance is not 0 or 1, or the
    // emission is only known after the walk): the factors of the two NEE terms apart, so that the shadow kernel can multiply them in the
    // reference's order -- (f*(T*e))/pdf*mis and ((T*e)*weight)*mis (TraceBase.cpp:144-174, 246-321) -- instead of (f*e/pdf*mis)*T
    A_NEE0,        // light sample: e.rgb (the light's emission towards the vertex, before the transmittance) | its pdf
    A_NEE1,        // bsdf sample: e.rgb | -
    A_NEE2,        // the two power-heuristic weights: light sample, bsdf sample | - | -
    A_COUNT
};

// The arrays sit in ONE allocation, addressed in GROUPS of eight: array a at poolg[a >> 3] + (a & 7)*stride.  Every group is below 4 GB
// (8 x 16 B x 2^25 slots at most), so a literal `a` costs one base pointer (an SGPR pair the compiler picks at compile time) and one 32-bit
// per-lane offset, and the pool as a whole can be larger than 4 GB (2^24 slots x (17 path + up to 20 walk arrays) = 9.9 GB).
#define PT_POOL_GROUP_SHIFT 3u
#define PT_POOL_GROUPS 5u                  /* (A_COUNT + 4 + (TGHIP_MAX_WIDE_DEPTH + 1)/2 + 7)/8 */
struct PathState {
    char * __restrict__ poolg[PT_POOL_GROUPS];   // base of arrays 8 g .. 8 g + 7
    uint32_t stride;                       // bytes per array
    // "pool_layout" = 1: slot records instead of arrays -- the eight arrays of a path's state (A_RAY_O .. A_SAMP, 128 bytes)
    // in ONE cache line per slot, the seven of its shadow-ray block (A_SH_O .. A_SH_P, 112 bytes) in a second one at
    // rec_shadow, the two auxiliary-output arrays and the three NEE-factor arrays at rec_aux: a sparse queue then touches one line per slot and block
    // instead of one partially used line per slot and ARRAY
    uint32_t records, rec_shadow, rec_aux;
    uint32_t * __restrict__ bm;            // queue bitmaps: queue q of workgroup b = words [q*bmStride + b*slots_per_block/32, ...)
    uint32_t bmStride;                     // words per queue
    float4 * __restrict__ partial;         // per work item: radiance sum, count (uint bits)
    BlockCtl * __restrict__ ctl;
    BlockStats * __restrict__ stats;
    uint32_t * __restrict__ live;          // [0] = tag of the last iteration that left work in some extension queue
    const uint32_t * __restrict__ abort_flag;   // non-zero: tghip_abort was called (one word that lives as long as the context, outside the pool)
    uint32_t num_slots, slots_per_block;
    uint32_t leaf_batch;                   // dynamic-fetch traversal: lanes waiting at a leaf before the leaf code runs (1 = at once)
    uint32_t leaf_batch_bvh2;              // the same for the BVH2 dynamic-fetch kernels (k_trace_closest_dyn / k_trace_shadow_dyn)
    // k_trace_closest_instw (instanced scenes, masters through the wide BVH): levels of the BVH2 stack (the scene's tree + the reference's tree over the
    // instances), lanes a phase of the turn needs to run next to a larger one, busy lanes at or below which the wave refills
    uint32_t inst_tree_depth, inst_phase_min, inst_refill_at;
    // Suspended walks ("walk time-slicing"): once a workgroup's queue has run dry, a wave of the wide traversal kernels that is
    // down to <= suspend_lanes busy lanes does not run its last, longest walks to the end at a few per cent lane occupancy -- it writes
    // the state of every walk that has had >= suspend_turns turns in this launch (WideState, the group stack, the best hit / the
    // shadow slot's partial result) to the slot's walk arrays, re-queues the slot and ends; the next launch of the same kernel picks the
    // walk up where it stopped, in a full wave.  A walk is a pure function of its ray, so hits and visit counts are unchanged.
    // suspend_lanes = 0: off.  Workgroups whose queue is shorter than suspend_min_queue never suspend (the end of a pass).
    uint32_t suspend_lanes, suspend_turns, suspend_min_queue;
    // The top of the 8-wide BVH -- nodes 0 .. lds_nodes-1 of the breadth-first array: the root, its children, ... -- is copied into LDS by the
    // DECOUPLED traversal kernels: every ray visits them, and what bounds the walk is the rate at which the CU's vector L1 takes lane
    // addresses (~1.1 16-byte lane-loads per clock, tools/ubench_chase.hip); ds_read_b128 does not go through it.  wide_depth: stack levels.
    uint32_t lds_nodes, wide_depth;
    uint32_t nee_factors;                  // the shading kernels leave the factors of the NEE terms apart (A_NEE0 .. A_NEE2) for k_trace_shadow<., FORWARD>
    uint32_t walk_base;                    // first of the walk arrays of the pool: +0 grpBase grpMasks triBase triMask, +1 triValid node sp -,
                                           // +2 (shadow slots) partial result.rgb | ray index, +3 tri2Base tri2Mask tri2Valid -,
                                           // +4.. the group stack, two 8-byte entries per array
};

// PT_POOL_RECORDS_RUNTIME: the record layout of the "pool_layout" experiment (-1 % on k_shade, +6 % on closest-hit: measured and not adopted) as a
// RUN-TIME option, as rounds 2-4 shipped it -- every slot access then carries the selects between two address computations and the two base
// pointers (branches and spilled scalars in the refill paths of every kernel).  Without the macro the arrays are the only layout.
#ifdef PT_POOL_RECORDS_RUNTIME
#define PT_RECORDS(st) ((st).records != 0u)
#else
#define PT_RECORDS(st) false
#endif
PT_DEV uint32_t slotOffset(const PathState &st, uint32_t a, uint32_t slot)     // `a` is a literal at every call site: the selects fold
{
    if (PT_RECORDS(st))
        return a < A_SH_O ? slot*128u + a*16u : a < A_AUX0 ? st.rec_shadow + slot*128u + (a - A_SH_O)*16u : st.rec_aux + slot*80u + (a - A_AUX0)*16u;
    return (a & ((1u << PT_POOL_GROUP_SHIFT) - 1u))*st.stride + slot*16u;
}
PT_DEV char *slotBase(const PathState &st, uint32_t a) { return PT_RECORDS(st) ? st.poolg[0] : st.poolg[a >> PT_POOL_GROUP_SHIFT]; }
// PT_NT_STATE (bit 0: loads, bit 1: stores, bit 2: the work items' partial sums and the samples' luminances too): the 16-byte accesses to the
// path pool carry the non-temporal hint (global_load / global_store ... nt).  A slot's state is written by one launch and read by the next,
// megabytes of other slots later; written as ordinary stores it is allocated in L2 on its way out and evicts the records, the attributes
// and the textures that the walks and the shading gathers do hit in.  Measured (profiles/r5_ab_nt_state.txt, two boxes, alternated):
// stores +5.2 % / +5.8 % on the metric's workload, +4 % on mesh1m, +13 % as shipped; LOADS -1 % alone and -3.5 % next to the stores (a slot's
// 16 bytes share their 128-byte line with the neighbouring slots other waves read, and a non-temporal load does not keep the line in L1).
// Hence 6: stores only.  The values moved are the same either way; 0 gives plain accesses back.
// WHERE: only the kernels that name NT explicitly -- the wavefront k_shade launches, the decoupled walks' bodies, finishBody / nextPath.
// Every other kernel (PT_NT_OTHER = 0) stores plainly, and must: with the hint in k_trace_shadow_wide -- one store, a finished slot's
// radiance, inside the walk's loop -- the two-level shadow walk of instanced scenes lost occluders in 48 % of the pixels of instances10k,
// differently in every run, and the sequential single-level walk (option decouple = 0) changed its image: the signature of round 3's loop-latch
// miscompile of that kernel (PT_TURN_JOIN), which the join had cured; without the hint the kernel is exact again
// (profiles/r5_nt_hazard.txt; tests/test_gpu_parity.py::test_instanced_shadow_walk_agrees_with_the_bvh2_walk and the scheduling test caught it).
#ifndef PT_NT_STATE
#define PT_NT_STATE 6
#endif
#ifndef PT_SLOT_EXTLOAD
#define PT_SLOT_EXTLOAD 1     /* the plain loads of the hinted kernels as ONE 16-byte vector load each (0: through float4, which the compiler narrows to
                                 global_load_dwordx3 where a kernel does not use the last word -- the kernels measured slower, wideRow below) */
#endif
#ifndef PT_NT_OTHER
#define PT_NT_OTHER 0         /* the kernels that do not pass NT explicitly (flat lists, BVH2 and two-level walks, the sequential wide walks, resolve) */
#endif
#ifndef PT_NT_TRAV
#define PT_NT_TRAV PT_NT_STATE   /* (bisecting aid: the hint in the traversal / finish kernels) */
#endif
#ifndef PT_NT_TAIL
#define PT_NT_TAIL 0          /* 1: k_tail's bodies keep the hint too (A/B) */
#endif
typedef float    PtF4v __attribute__((ext_vector_type(4)));
typedef uint32_t PtU4v __attribute__((ext_vector_type(4)));
// (NT: the PT_NT_STATE bits in force at the call site -- the kernels whose slots are re-read within microseconds, the fused flat-list
// launches and k_tail, pass 0: there the hint costs 1.3 %, profiles/r5_ab_nt_state.txt)
// (the plain halves go through float4 / uint4 themselves, as the references of rounds 1-4 did: the same code as then where NT = 0)
template<int NT> struct SlotF4Ref {
    PtF4v *p;
    PT_DEV operator float4() const
    {
        if constexpr ((NT & 1) != 0) { const PtF4v v = __builtin_nontemporal_load(p); return make_float4(v.x, v.y, v.z, v.w); }
        else if constexpr (NT != 0 && PT_SLOT_EXTLOAD) { const PtF4v v = *p; return make_float4(v.x, v.y, v.z, v.w); }
        else return *reinterpret_cast<const float4 *>(p);
    }
    PT_DEV const SlotF4Ref &operator=(float4 v) const
    {
        if constexpr ((NT & 2) != 0) { const PtF4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, p); }
        else *reinterpret_cast<float4 *>(p) = v;
        return *this;
    }
};
template<int NT> struct SlotU4Ref {
    PtU4v *p;
    PT_DEV operator uint4() const
    {
        if constexpr ((NT & 1) != 0) { const PtU4v v = __builtin_nontemporal_load(p); return make_uint4(v.x, v.y, v.z, v.w); }
        else if constexpr (NT != 0 && PT_SLOT_EXTLOAD) { const PtU4v v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
        else return *reinterpret_cast<const uint4 *>(p);
    }
    PT_DEV const SlotU4Ref &operator=(uint4 v) const
    {
        if constexpr ((NT & 2) != 0) { const PtU4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, p); }
        else *reinterpret_cast<uint4 *>(p) = v;
        return *this;
    }
};
template<int NT = PT_NT_OTHER>
PT_DEV SlotF4Ref<NT> slotF4(const PathState &st, uint32_t a, uint32_t slot)
{
    return SlotF4Ref<NT>{reinterpret_cast<PtF4v *>(slotBase(st, a) + (size_t)slotOffset(st, a, slot))};
}
template<int NT = PT_NT_OTHER>
PT_DEV SlotU4Ref<NT> slotU4(const PathState &st, uint32_t a, uint32_t slot)
{
    return SlotU4Ref<NT>{reinterpret_cast<PtU4v *>(slotBase(st, a) + (size_t)slotOffset(st, a, slot))};
}
// one 32-bit word of a slot's 16 bytes (the few places that touch less than the whole vector)
PT_DEV float &slotW(const PathState &st, uint32_t a, uint32_t slot, uint32_t word)
{
    return *reinterpret_cast<float *>(slotBase(st, a) + (size_t)slotOffset(st, a, slot) + word*4u);
}
PT_DEV uint32_t &slotUW(const PathState &st, uint32_t a, uint32_t slot, uint32_t word)
{
    return *reinterpret_cast<uint32_t *>(slotBase(st, a) + (size_t)slotOffset(st, a, slot) + word*4u);
}


#define PT_PASS_THINLENS 0x100u   /* internal pass flag: the scene's camera is a thin lens (nextPath's EXT variants sample the lens) */
#define PT_PASS_MEDIA    0x200u   /* internal pass flag: the scene has participating media (new paths start in the camera's medium) */
struct PassParams {
    uint32_t spp_begin, spp_end, seed;
    uint32_t chunk;            // samples per work item
    uint32_t chunks;           // items per pixel slot = ceil((spp_end - spp_begin)/chunk)
    uint32_t group;            // k_resolve: partial sums added up in groups of this many before they reach the pixel (4 for one-sample items:
                               // the sum then rounds exactly like that of four-sample items, whatever the shard or batch -- tungsten_hip.hip)
    uint32_t pix_slots;        // pixel slots in this batch (= tiles in batch * 256)
    uint32_t total_items;      // pix_slots*chunks (of this launch's workgroups: one half of the batch when the pool runs as two halves)
    uint32_t item_begin;       // first item of this launch's workgroups within the batch
    uint32_t first_tile;       // first owned tile of the batch (index into the shard's tile list)
    uint32_t shard_index, shard_count;
    uint32_t shard_skew;       // tghip_shard_skew(shard_count): tile (tx, ty) belongs to shard (tx + ty*skew) % shard_count
    const uint32_t *owned_tiles;   // the shard's tiles in row-major order (nullptr when shard_count == 1: every tile, identity)
    uint32_t tiles_x, num_tiles;
    uint32_t width, height;
    uint32_t iter_tag;         // wavefront iteration number (liveness reporting of the fused flat-scene kernels)
    // ---- TGHIP_PASS_SOBOL / TGHIP_PASS_RECORDS passes (read only by the kernels' `EXT` code paths) ----
    uint32_t flags;            // TGHIP_PASS_* | PT_PASS_*
    uint32_t variance_w;       // SampleRecords per image row = ceil(width/4)
    const uint32_t *tile_seeds;   // SobolPathSampler seed per 16x16 tile (device copy of TgHipPassDesc::tile_seeds)
    // per SampleRecord (nullptr unless TGHIP_PASS_RECORDS): first sample index, samples per pixel this pass, and the
    // offset of the record's 16 x count block in `lum`; spp_begin/spp_end are then RELATIVE to the record's first index
    const uint32_t *rec_index, *rec_count, *rec_lum;
    // Work items of a record pass are enumerated without gaps: the owned records are sorted by descending sample count,
    // so the pixels that still have samples in chunk c are a PREFIX of the sorted pixel list (16 per record).  Item
    // w (+ item_base) -> chunk c with rec_chunk_start[c] <= w < rec_chunk_start[c + 1] (rec_hint[w/64] = c of item
    // 64*(w/64)), pixel slot j = w - rec_chunk_start[c] -> record rec_sorted[j/16], pixel j%16 of it.
    const uint32_t *rec_sorted, *rec_chunk_start, *rec_hint;
    uint32_t item_base;        // first item of this batch in the pass's enumeration
    uint32_t num_sorted;       // owned records
    uint32_t num_chunks;       // chunks of the record with the most samples
    float *lum;                // luminance of every sample of the pass, in the order SampleRecord::addSample saw them
    TgHipAuxPixel *aux;        // TGHIP_PASS_AUX: the auxiliary output buffers, one record per image pixel
    float *samples;            // TGHIP_PASS_SAMPLES: radiance of every sample of the pass, [pixel][sample - samples_begin][rgb]
    uint32_t samples_begin, samples_spp;
};

// OutputBuffer<T>::addSample (cameras/OutputBuffer.hpp:104-132) with _bufferB and _variance present, on the channels
// [ch0, ch0 + N) of one pixel.  A TGHIP_PASS_AUX pass hands all samples of a pixel to ONE slot in index order (chunk = the
// whole pass), and passes follow each other, so every pixel sees its samples in the reference's order.
PT_DEV void auxAdd3(TgHipAuxPixel &px, int output, int ch0, int n, float c0, float c1, float c2)   // n = 1 or 3 channels
{
    if (isnan(c0) || isinf(c0) || (n == 3 && (isnan(c1) || isinf(c1) || isnan(c2) || isinf(c2))))
        return;
    const uint32_t sampleIdx = px.count[output]++;
    for (int k = 0; k < n; ++k) {
        const float c = k == 0 ? c0 : k == 1 ? c1 : c2;
        float a = px.a[ch0 + k], b = px.b[ch0 + k];
        float curr = a;
        if (sampleIdx > 0) {
            uint32_t sampleCountA = (sampleIdx + 1u)/2u, sampleCountB = sampleIdx/2u;
            curr = (a*(float)sampleCountA + b*(float)sampleCountB)/(float)sampleIdx;
        }
        float delta = c - curr;
        curr += delta/(float)(sampleIdx + 1u);
        px.variance[ch0 + k] += delta*(c - curr);
        const uint32_t perBufferSampleCount = sampleIdx/2u + 1u;
        if (sampleIdx & 1u) px.b[ch0 + k] = b + (c - b)/(float)perBufferSampleCount;
        else                px.a[ch0 + k] = a + (c - a)/(float)perBufferSampleCount;
    }
}

PT_DEV uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ---- small scene tables in LDS --------------------------------------------------------------------
// The shading kernels chase object -> bsdf -> texture -> light records per lane.  Those tables are tiny, but the
// vector L1 is flushed continuously by the streaming path state, so every dependent lookup pays L2 latency.
// Each workgroup copies them into LDS once and the lookups become LDS reads (the big arrays -- records, attributes,
// texels, CDFs -- stay in global memory).
// Round 5: the copy is UNCONDITIONAL.  With the run-time fallback "too large: keep the global tables" every table pointer was a select
// of an LDS and a global address, so the compiler could not infer the address space and every lookup became a flat_load -- 315 of them in
// the class-0 variant, each behind an `s_waitcnt vmcnt(0) lgkmcnt(0)` that also drains every global load in flight (235 such waits).
// Whether the tables fit is decided by the host at upload (tungsten_hip.hip: tablesFit, the same arithmetic as below); scenes whose
// tables do not fit shade with the GLOBAL_TABLES instantiation of the all-features variant, which does not stage at all.  The sampled
// environment map's marginal tables come along when env_tex >= 0 (the host clears env_tex when they do not fit next to the rest).
#define PT_LDS_TABLE_BYTES 12288u
struct SceneTableLayout { uint32_t offBsdf, offTex, offLights, offEnv, offEnvG, total; };
__host__ __device__ inline SceneTableLayout sceneTableLayout(uint32_t numObjects, uint32_t numBsdfs, uint32_t numTextures, uint32_t numLights, uint32_t numInfinite, int envH)
{
    SceneTableLayout l;
    const uint32_t szObj = numObjects*(uint32_t)sizeof(TgHipObject), szBsdf = numBsdfs*(uint32_t)sizeof(TgHipBsdf), szTex = numTextures*(uint32_t)sizeof(TgHipTexture);
    const uint32_t szLights = (numLights + numInfinite)*(uint32_t)sizeof(int32_t);
    l.offBsdf = (szObj + 15u) & ~15u;
    l.offTex = (l.offBsdf + szBsdf + 15u) & ~15u;
    l.offLights = (l.offTex + szTex + 15u) & ~15u;
    // the marginal tables of the sampled environment map (mpdf[h] mcdf[h + 1], then its 513-entry guide): the head of the
    // envmap-sampling chain becomes LDS reads
    l.offEnv = (l.offLights + szLights + 15u) & ~15u;
    const uint32_t szEnvF = envH > 0 ? (2u*(uint32_t)envH + 1u)*4u : 0u, szEnvG = envH > 0 ? (PT_GUIDE_MARGINAL + 1u)*2u : 0u;
    l.offEnvG = (l.offEnv + szEnvF + 3u) & ~3u;
    l.total = envH > 0 ? ((l.offEnvG + szEnvG + 3u) & ~3u) : l.offLights + szLights;
    return l;
}
PT_DEV DeviceScene stageSceneTables(const DeviceScene &s, unsigned char *lds)
{
    const bool env = s.env_tex >= 0;
    const SceneTableLayout l = sceneTableLayout(s.num_objects, s.num_bsdfs, s.num_textures, s.num_lights, s.num_infinite_lights, env ? s.env_h : 0);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    auto copy = [&](uint32_t off, const void *src, uint32_t bytes) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(src);
        for (uint32_t i = threadIdx.x; i < bytes/4u; i += blockDim.x)
            dst[off/4u + i] = w[i];
    };
    copy(0, s.objects, s.num_objects*(uint32_t)sizeof(TgHipObject));
    copy(l.offBsdf, s.bsdfs, s.num_bsdfs*(uint32_t)sizeof(TgHipBsdf));
    copy(l.offTex, s.textures, s.num_textures*(uint32_t)sizeof(TgHipTexture));
    copy(l.offLights, s.lights, s.num_lights*(uint32_t)sizeof(int32_t));
    copy(l.offLights + s.num_lights*(uint32_t)sizeof(int32_t), s.infinite_lights, s.num_infinite_lights*(uint32_t)sizeof(int32_t));
    if (env) {
        copy(l.offEnv, s.env_marginal, (2u*(uint32_t)s.env_h + 1u)*4u);
        copy(l.offEnvG, s.env_guide, ((PT_GUIDE_MARGINAL + 1u)*2u + 3u) & ~3u);     // (the guide array is padded to a multiple of 4 bytes at upload)
    }
    __syncthreads();
    DeviceScene r = s;
    // (unconditionally LDS addresses: without a sampled environment map nothing reads the two env pointers)
    r.env_marginal = reinterpret_cast<const float *>(lds + l.offEnv);
    r.env_guide = reinterpret_cast<const uint16_t *>(lds + l.offEnvG);
    r.objects = reinterpret_cast<const TgHipObject *>(lds);
    r.bsdfs = reinterpret_cast<const TgHipBsdf *>(lds + l.offBsdf);
    r.textures = reinterpret_cast<const TgHipTexture *>(lds + l.offTex);
    r.lights = reinterpret_cast<const int32_t *>(lds + l.offLights);
    r.infinite_lights = r.lights + s.num_lights;
    return r;
}

// Workgroup-local state staged in LDS for the duration of one kernel.
template<uint32_t WORDS>
struct BlockLdsT {
    uint32_t bm[Q_COUNT][WORDS];            // queue bitmaps being built / consumed
    uint32_t prefix[WORDS];
    uint32_t n;                                      // length of the consumed queue
    uint32_t cursor;
    uint32_t samples, closest_rays, shadow_rays, shadow_slots, nodes, prims;
};
typedef BlockLdsT<PT_MAX_WORDS> BlockLds;
// the static-fetch and BVH2 traversal kernels (flat-list and instanced scenes, wide BVH switched off) run with <= 2048 slots per
// workgroup (slotCap in the shim): their deep stacks need the LDS
typedef BlockLdsT<64u> BlockLdsSmall;

// queue of the paths whose hit is of shading class `cls` (CLS_MISS: no hit)
PT_DEV int shadeQueue(int cls) { return cls == 0 ? Q_SHADE0 : cls == 1 ? Q_SHADE1 : cls == 2 ? Q_SHADE2 : cls == 3 ? Q_SHADE3 : Q_MISS; }

// push: set the slot's bit in queue q (LDS atomic OR)
template<class LDS>
PT_DEV void queuePush(bool push, uint32_t localSlot, LDS &L, int q)
{
    if (push)
        atomicOr(&L.bm[q][localSlot >> 5], 1u << (localSlot & 31u));
}

// Kernel prologue: loads the bitmaps this kernel appends to (bit mask `appendMask` over Q_*), expands the
// consumed queue `q` (-1 = none), followed by the optional second consumed queue `q2`, into order[0 .. L.n)
// (ascending local slot indices per queue; `order` = LDS scratch of slots_per_block entries), and resets the
// statistics.  All threads must call it.
// Expands queue q, followed by the optional queue q2, from the LDS bitmaps into order[0 .. L.n) and clears them
// (consumed).  All threads must call it; it ends with a barrier.
template<class LDS>
PT_DEV void queuesExpand(LDS &L, const PathState &st, int q, int q2, unsigned short *order)
{
    const uint32_t W = st.slots_per_block >> 5;
    const uint32_t t = threadIdx.x;
    if (t == 0) L.n = 0;
    uint32_t done = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int qq = pass == 0 ? q : q2;
        if (qq < 0)
            continue;                            // uniform
        // exclusive prefix of the per-word popcounts: the first wave, 64 words at a time
        if (t < 64) {
            uint32_t run = done;
            for (uint32_t base = 0; base < W; base += 64u) {
                const uint32_t wd = base + t;
                uint32_t c = wd < W ? (uint32_t)__popc(L.bm[qq][wd]) : 0u;
                uint32_t inc = c;
                for (int off = 1; off < 64; off <<= 1) {
                    uint32_t v = __shfl_up(inc, off);
                    if ((int)t >= off) inc += v;
                }
                if (wd < W) L.prefix[wd] = run + inc - c;
                run += __shfl(inc, 63);
            }
            if (t == 63) L.n = run;
        }
        __syncthreads();
        for (uint32_t wd = t; wd < W; wd += blockDim.x) {
            uint32_t bits = L.bm[qq][wd], off = L.prefix[wd];
            while (bits) {
                uint32_t b = (uint32_t)__ffs((int)bits) - 1u;
                order[off++] = (unsigned short)(wd*32u + b);
                bits &= bits - 1u;
            }
            L.bm[qq][wd] = 0u;                   // consumed
        }
        __syncthreads();
        done = L.n;
    }
}

// freshAppend: the appended bitmaps start empty and queuesEnd(atomicAppend) ORs them into the global ones -- for launches that run
// concurrently with others appending to the same workgroup's queues (the shading classes, each consuming a queue of its own).
template<class LDS>
PT_DEV void queuesBegin(LDS &L, const PathState &st, const BlockCtl &ctl, int q, uint32_t appendMask, unsigned short *order,
                        int q2 = -1, bool freshAppend = false)
{
    const uint32_t W = st.slots_per_block >> 5;
    const uint32_t t = threadIdx.x;
    for (uint32_t wd = t; wd < W; wd += blockDim.x) {
#pragma unroll
        for (int k = 0; k < Q_COUNT; ++k) {
            bool load = k == q || k == q2 || (!freshAppend && ((appendMask >> k) & 1u));
            L.bm[k][wd] = load ? st.bm[(uint32_t)k*st.bmStride + blockIdx.x*W + wd] : 0u;
        }
    }
    if (t == 0) {
        L.cursor = ctl.item_cursor;
        L.samples = L.closest_rays = L.shadow_rays = L.shadow_slots = L.nodes = L.prims = 0;
        L.n = 0;
    }
    __syncthreads();
    queuesExpand(L, st, q, q2, order);
}

// A thread's entries order[k*blockDim + tid], k < 16, packed two per register, so that the LDS scratch holding
// `order` can be reused (as the traversal stack) once every thread has fetched its share (needs
// slots_per_block <= 16*blockDim).
struct OrderRegs { uint32_t p[8]; };
PT_DEV OrderRegs orderPreload(const unsigned short *order, uint32_t n)
{
    OrderRegs r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t i0 = (uint32_t)(2*k)*blockDim.x + threadIdx.x, i1 = (uint32_t)(2*k + 1)*blockDim.x + threadIdx.x;
        uint32_t lo = i0 < n ? order[i0] : 0u, hi = i1 < n ? order[i1] : 0u;
        r.p[k] = lo | (hi << 16);
    }
    __syncthreads();
    return r;
}
PT_DEV uint32_t orderGet(const OrderRegs &r, uint32_t k)   // k is wave-uniform
{
    uint32_t h = k >> 1, w = r.p[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) w = h == (uint32_t)j ? r.p[j] : w;
    return (k & 1u) ? (w >> 16) : (w & 0xFFFFu);
}

// Kernel epilogue: writes back the consumed (now empty) and appended bitmaps.  Returns (to thread 0..W-1) nothing;
// `anyExt` tells whether the extension queue holds work.
template<class LDS>
PT_DEV bool queuesEnd(LDS &L, const PathState &st, int q, uint32_t appendMask, int q2 = -1, bool atomicAppend = false, uint32_t liveMask = 0u)
{
    __syncthreads();
    const uint32_t W = st.slots_per_block >> 5;
    const uint32_t t = threadIdx.x;
    uint32_t ext = 0;
    for (uint32_t wd = t; wd < W; wd += blockDim.x) {
#pragma unroll
        for (int k = 0; k < Q_COUNT; ++k) {
            if (k == q || k == q2 || (!atomicAppend && ((appendMask >> k) & 1u)))
                st.bm[(uint32_t)k*st.bmStride + blockIdx.x*W + wd] = L.bm[k][wd];
            else if (atomicAppend && ((appendMask >> k) & 1u) && L.bm[k][wd])
                atomicOr(&st.bm[(uint32_t)k*st.bmStride + blockIdx.x*W + wd], L.bm[k][wd]);
        }
        ext |= L.bm[Q_EXT][wd] | L.bm[Q_EXTP][wd];
#pragma unroll
        for (int k = 0; k < Q_COUNT; ++k)
            if ((liveMask >> k) & 1u) ext |= L.bm[k][wd];     // (only bitmaps this kernel loaded: q, q2 or non-atomic appendMask)
    }
    return __syncthreads_or(ext != 0u) != 0;
}

// pixel slot j of the batch -> image pixel (16x16 tile dicing, PathTraceIntegrator.cpp:27-42; a shard owns the tiles
// tghip_tile_owner deals it, listed in owned_tiles).  False for slots of an edge tile that fall outside the image.
PT_DEV bool slotPixel(const PassParams &pp, uint32_t j, uint32_t &x, uint32_t &y)
{
    uint32_t tileLocal = j >> 8, inTile = j & 255u;
    uint32_t k = pp.first_tile + tileLocal;          // k-th owned tile (the batch never reaches past the shard's list)
    uint32_t tile = pp.owned_tiles ? at32(pp.owned_tiles, k) : k;
    if (tile >= pp.num_tiles)
        return false;
    uint32_t tx = tile % pp.tiles_x, ty = tile/pp.tiles_x;
    x = tx*16u + (inTile & 15u);
    y = ty*16u + (inTile >> 4);
    return x < pp.width && y < pp.height;
}

// PinholeCamera::sampleDirection + ReconstructionFilter::sample (PinholeCamera.cpp:70-86,
// ReconstructionFilter.hpp:86-103,152-169).  Consumes two random numbers.
typedef const PT_CONST_AS TgHipCamera &CameraRef;   // constant address space: uniform loads become s_load
PT_DEV float filterSample1D(CameraRef cam, float xi)
{
    bool negative = xi < 0.5f;
    xi = negative ? xi*2.0f : (xi - 0.5f)*2.0f;
    // first i in [0, 30) with xi < cdf[i], else 30 (cdf[0] = 0, so i >= 1).  Walking downwards with selects keeps
    // every table index a compile-time constant (the table sits in SGPRs; a per-lane index would not).
    int idx = 30;
    float hi = cam.filter_cdf[30], lo = cam.filter_cdf[29];
#pragma unroll
    for (int i = 29; i >= 1; --i) {
        bool below = xi < cam.filter_cdf[i];
        idx = below ? i : idx;
        hi = below ? cam.filter_cdf[i] : hi;
        lo = below ? cam.filter_cdf[i - 1] : lo;
    }
    float pdf = hi - lo;
    float u = cam.filter_bin_size*(idx + (xi - lo)/pdf);
    return negative ? -u : u;
}
// l0, l1: the lens sample a thin-lens camera draws first (ThinlensCamera::samplePosition, cameras/ThinlensCamera.cpp:85-97,
// default DiskTexture aperture); xi0, xi1: the pixel-filter sample.  False when the direction sample fails (cat-eye
// vignetting, :119-124): PathTracer::traceSample then returns black for the sample (PathTracer.cpp:27-28).
// `dist`: the scene's Distribution2D tables (a bitmap aperture's live there, TgHipCamera::aperture_dist)
template<bool LENS>
PT_DEV bool cameraRay(CameraRef cam, bool lens, uint32_t px, uint32_t py, float l0, float l1, float xi0, float xi1, f3 &o, f3 &d, const float *dist = nullptr)
{
    float fu = 0.0f, fv = 0.0f;
    if (cam.filter_type == TGHIP_FILTER_BOX) { fu = xi0 - 0.5f; fv = xi1 - 0.5f; }
    else if (cam.filter_type == TGHIP_FILTER_TABULATED) { fu = filterSample1D(cam, xi0); fv = filterSample1D(cam, xi1); }
    if (LENS && lens) {
        float su, sv;                                                   // _aperture->sample(MAP_UNIFORM, lensUv)
        if (cam.aperture_type == TGHIP_APERTURE_BLADE) {                // BladeTexture::sample (textures/BladeTexture.cpp:110-130)
            float u = l0*(float)cam.blade_count;
            int blade = (int)u;
            u -= (float)blade;
            float phi = cam.blade_angle + (float)blade*cam.blade_step;
            float sinPhi, cosPhi;
            sincosfH(phi, sinPhi, cosPhi);
            float uSqrt = sqrtf(u);
            float alpha = 1.0f - uSqrt, beta = (1.0f - l1)*uSqrt;
            float lx = (1.0f + cam.blade_edge[0])*beta + (1.0f - alpha - beta), ly = cam.blade_edge[1]*beta;
            su = (lx*cosPhi - ly*sinPhi)*0.5f + 0.5f;
            sv = (ly*cosPhi + lx*sinPhi)*0.5f + 0.5f;
        } else if (cam.aperture_type == TGHIP_APERTURE_BITMAP) {
            // BitmapTexture::sample(MAP_UNIFORM, uv) (textures/BitmapTexture.cpp:433-439) = Distribution2D::warp (sampling/Distribution2D.hpp:68-77):
            // the row by the second number, the column by the first, each followed by the position inside the texel
            const int w = cam.aperture_w, h = cam.aperture_h;
            const float *mpdf = dist + cam.aperture_dist, *mcdf = mpdf + h, *pdf = mcdf + h + 1, *cdf = pdf + (size_t)w*h;
            int lo = 0, hi = h + 1;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (mcdf[mid] <= l1) lo = mid + 1; else hi = mid; }
            const int row = min(max(lo - 1, 0), h - 1);                  // (a table that is not a CDF -- NaN, all zero -- must not index outside itself)
            const float nv = fminf(fmaxf((l1 - mcdf[row])/mpdf[row], 0.0f), 1.0f);
            const float *rowCdf = cdf + (size_t)row*(w + 1);
            lo = 0; hi = w + 1;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (rowCdf[mid] <= l0) lo = mid + 1; else hi = mid; }
            const int column = min(max(lo - 1, 0), w - 1);
            const float nu = fminf(fmaxf((l0 - rowCdf[column])/pdf[(size_t)row*w + column], 0.0f), 1.0f);
            su = (nu + (float)column)/(float)w;
            sv = 1.0f - (nv + (float)row)/(float)h;
        } else {
            float phi = l0*PT_TWO_PI, r = sqrtf(l1);                   // SampleWarp::uniformDisk (SampleWarp.hpp:64-69)
            float sinPhi, cosPhi;
            sincosfH(phi, sinPhi, cosPhi);
            su = cosPhi*r*0.5f + 0.5f;
            sv = sinPhi*r*0.5f + 0.5f;
        }
        float ax = (su*2.0f - 1.0f)*cam.aperture_size;
        float ay = (sv*2.0f - 1.0f)*cam.aperture_size;
        o = mk3(cam.xf[0]*ax + cam.xf[1]*ay + cam.xf[2]*0.0f + cam.pos[0],
                cam.xf[3]*ax + cam.xf[4]*ay + cam.xf[5]*0.0f + cam.pos[1],
                cam.xf[6]*ax + cam.xf[7]*ay + cam.xf[8]*0.0f + cam.pos[2]);
        f3 planePos = mk3(-1.0f + ((float)px + fu)*2.0f*cam.pixel_size_x,       // no + 0.5 here (ThinlensCamera.cpp:110-114)
                          cam.ratio - ((float)py + fv)*2.0f*cam.pixel_size_x,
                          cam.plane_dist);
        planePos = planePos*(cam.focus_dist/planePos.z);
        f3 lensPos = mk3(cam.inv_xf[0]*o.x + cam.inv_xf[1]*o.y + cam.inv_xf[2]*o.z + cam.inv_xf[3],
                         cam.inv_xf[4]*o.x + cam.inv_xf[5]*o.y + cam.inv_xf[6]*o.z + cam.inv_xf[7],
                         cam.inv_xf[8]*o.x + cam.inv_xf[9]*o.y + cam.inv_xf[10]*o.z + cam.inv_xf[11]);
        f3 localD = normalized(planePos - lensPos);
        d = mat3Mul(cam.xf, localD);
        if (cam.cat_eye > 0.0f) {
            float k = cam.cat_eye*cam.plane_dist;
            float dx = lensPos.x - k*localD.x/localD.z, dy = lensPos.y - k*localD.y/localD.z;
            if (dx*dx + dy*dy > cam.aperture_size*cam.aperture_size)
                return false;
        }
        return true;
    }
    f3 localD = normalized(mk3(-1.0f + ((float)px + 0.5f + fu)*2.0f*cam.pixel_size_x,
                               cam.ratio - ((float)py + 0.5f + fv)*2.0f*cam.pixel_size_x,
                               cam.plane_dist));
    o = ld3(cam.pos);
    d = mat3Mul(cam.xf, localD);
    return true;
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

// ---- flat lists walked as the reference's tree -----------------------------------------------
// The reference intersects a scene through Embree's BVH4 over its finite primitives, one per leaf, and with coincident faces (a block
// standing ON the floor quad, a light lying IN the ceiling, a ray into the seam of two walls) Embree's visiting rules decide which primitive
// a ray hits (include/tungsten_hip.h: TgHipTopNode has the rules and the citations; oracle.c: embree_top_walk is the test side's copy).
// Scenes that carry the tree (DeviceScene::top_nodes: flat lists of analytic primitives) are intersected by flatClosestOrdered.
struct EmbreeRay { float o[3], rdir[3], tNear, tFar; };
PT_DEV EmbreeRay embreeRay(const RayD &ray)
{
    EmbreeRay e;
    const float d[3] = {ray.d.x, ray.d.y, ray.d.z};
    e.o[0] = ray.o.x; e.o[1] = ray.o.y; e.o[2] = ray.o.z;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        e.rdir[k] = embreeRcp(fabsf(d[k]) < 1e-18f ? 1e-18f : d[k]);
    e.tNear = fmaxf(ray.tmin, 0.0f); e.tFar = fmaxf(ray.tmax, 0.0f);
    return e;
}
// the node's slab test for one child box (embreeBoxVisible's arithmetic) under e.tFar; entry = the box's entry distance
PT_DEV bool embreeLeafEntry(const EmbreeRay &e, const float *lo, const float *hi, float &entry)
{
    int n[3], f[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool pos = e.rdir[k] >= 0.0f;
        n[k] = __float_as_int(((pos ? lo[k] : hi[k]) - e.o[k])*e.rdir[k]);
        f[k] = __float_as_int(((pos ? hi[k] : lo[k]) - e.o[k])*e.rdir[k]);
    }
    const int nn = max(max(n[0], n[1]), max(n[2], __float_as_int(e.tNear)));     // maxi / mini: the bit patterns as signed integers
    const int ff = min(min(f[0], f[1]), min(f[2], __float_as_int(e.tFar)));
    entry = __int_as_float(nn);
    return !(nn > ff);
}
// BVH4Intersector1::intersect over the tree, in full (bvh_intersector1.cpp:60-125): runs for the few rays flatClosestOrdered cannot decide
// from the plain list.  The stack holds (child, entry distance as its bit pattern); TOP_STACK bounds what a tree of TGHIP_TOP_MAX_DEPTH levels
// can leave on it (three waiting children per level) -- tghip_upload_scene refuses deeper trees.
// NOT inlined: one ray in a few thousand comes here, and inlined the walk's stack and sorting networks cost every variant of the fused
// shading kernels registers (and the 168-register variants hundreds of spills) on the path every ray takes.  The function gets what it reads
// as plain pointers -- a reference to the kernel's DeviceScene would have to be materialised in memory for the call.
#define TOP_STACK 24
template<uint32_t KINDS>
__device__ __attribute__((noinline)) float4 flatOrderedWalk(const float *topNodes, const float4 *recs, const TgHipObject *objects,
                                                            float ox, float oy, float oz, float tmin, float dx, float dy, float dz, float tmax)
{
    DeviceScene s;                                      // (only what testRecord reads)
    s.top_nodes = topNodes; s.recs = recs; s.objects = objects;
    RayD ray;                                           // ray.tmax is Embree's ray.tfar
    ray.o = mk3(ox, oy, oz); ray.d = mk3(dx, dy, dz); ray.tmin = tmin; ray.tmax = tmax;
    EmbreeRay e = embreeRay(ray);
    float4 hit = make_float4(ray.tmax, 0.0f, 0.0f, __int_as_float(-1));
    int sref[TOP_STACK];
    uint32_t sdist[TOP_STACK];
    int sp = 0;
    sref[0] = 0; sdist[0] = 0xFF800000u; sp = 1;       // the root, dist = neg_inf
    while (sp > 0) {
        --sp;
        int cur = sref[sp];
        if (__uint_as_float(sdist[sp]) > ray.tmax) continue;           // popped behind the hit so far
        bool descend = true;
        while (cur >= 0) {
            const float *n = s.top_nodes + (uint32_t)cur*28u;
            int c[4]; uint32_t d[4]; int nh = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int child = __float_as_int(n[24 + i]);
                float entry;
                const bool in = child != TGHIP_TOP_EMPTY && embreeLeafEntry(e, n + 3*i, n + 12 + 3*i, entry);
                if (in) { c[nh] = child; d[nh] = __float_as_uint(entry); nh++; }
            }
            if (nh == 0) { descend = false; break; }
            if (nh == 1) { cur = c[0]; continue; }
            if (nh == 2) {
                if (d[0] < d[1]) { sref[sp] = c[1]; sdist[sp] = d[1]; sp++; cur = c[0]; }
                else             { sref[sp] = c[0]; sdist[sp] = d[0]; sp++; cur = c[1]; }
                continue;
            }
            // three / four children: pushed in slot order, sorted by Embree's networks (common/stack_item.h:44-60; s1 = the top of the stack)
            int r1, r2, r3, r4 = 0; uint32_t d1, d2, d3, d4 = 0;
            auto sw = [](int &ra, uint32_t &da, int &rb, uint32_t &db) { const int r = ra; ra = rb; rb = r; const uint32_t x = da; da = db; db = x; };
            if (nh == 3) {
                r1 = c[2]; d1 = d[2]; r2 = c[1]; d2 = d[1]; r3 = c[0]; d3 = d[0];
                if (d2 < d1) sw(r2, d2, r1, d1);
                if (d3 < d2) sw(r3, d3, r2, d2);
                if (d2 < d1) sw(r2, d2, r1, d1);
                sref[sp] = r3; sdist[sp] = d3; sref[sp + 1] = r2; sdist[sp + 1] = d2; sp += 2;
            } else {
                r1 = c[3]; d1 = d[3]; r2 = c[2]; d2 = d[2]; r3 = c[1]; d3 = d[1]; r4 = c[0]; d4 = d[0];
                if (d2 < d1) sw(r2, d2, r1, d1);
                if (d4 < d3) sw(r4, d4, r3, d3);
                if (d3 < d1) sw(r3, d3, r1, d1);
                if (d4 < d2) sw(r4, d4, r2, d2);
                if (d3 < d2) sw(r3, d3, r2, d2);
                sref[sp] = r4; sdist[sp] = d4; sref[sp + 1] = r3; sdist[sp + 1] = d3; sref[sp + 2] = r2; sdist[sp + 2] = d2; sp += 3;
            }
            cur = r1;
        }
        if (!descend) continue;
        testRecord<false, KINDS>(s, (uint32_t)~cur, ray, ray.tmax, hit);
        e.tFar = ray.tmax;                                                // ray_far = ray.tfar
    }
    return hit;
}
// Every lane tests the whole list against the ray's own tmax (uniform record loads, as the plain walk) and keeps, of the records it hits
// AND whose leaf box it passes (a leaf whose box the ray misses under its own tmax is never reached by the walk: the slab test only gets
// stricter as the hit distance shrinks -- the light a shadow ray ends on is the usual case, its flat box an ulp behind tmax), the nearest hit
// b and the distance t2 of the second nearest.  The walk returns b whenever b is the strict minimum and its box is not entered behind t2: a
// box contains its children's, so b's ancestors are passed and popped no later than b; no hit before b's turn can make b's pop fail (the hits
// so far are >= t2 >= entry); b's own test accepts t_b under any of them (t_b < t2); nothing behind t_b is accepted afterwards -- whatever
// the tree looks like.  With no such record the walk finds nothing.  Otherwise (ties, a box entered an ulp behind another primitive's hit)
// the lane walks the tree.
template<bool COUNT, uint32_t KINDS>
PT_DEV float4 flatClosestOrdered(const DeviceScene &s, const RayD &ray, uint32_t &primsTested)
{
    // Round 5: the slab test is out of the per-record loop.  The loop keeps the nearest hit b (the first of equal ones), the distance t2 of the
    // second nearest hit -- whatever the boxes say: a lower bound of the second nearest hit the walk can reach, which only makes the rule
    // stricter --, the number of records hit, and whether a hit has a NaN distance (a ray IN a disk's plane: Disk::intersect divides 0 by 0 and
    // accepts it, as the reference does; such a hit cannot be ordered).  ONE slab test afterwards, of b's leaf box:
    //   nothing hit                          -> nothing
    //   b's box passed, entered at `entry`   -> b when t_b < t2 strictly and entry <= t2 (the argument of DESIGN.md 4f), else the walk
    //   b's box missed                       -> nothing when b is the only record hit (the walk cannot reach it); else the walk
    //   a NaN distance                       -> the walk
    // (oracle.c: flat_shortcut_decides is this rule; tests/test_flat_order.py holds it to the walk on rays made to tie.  The number of records
    // hit is not counted: none <=> no hit kept, one <=> t2 still infinite -- a hit at a NaN distance leaves t2 alone, but then the lane walks)
    const uint32_t n = s.num_recs;
    float4 hit = make_float4(ray.tmax, 0.0f, 0.0f, __int_as_float(-1));
    float tb = PT_INF, t2 = PT_INF;
    bool unordered = false;
    for (uint32_t i = 0; i < n; ++i) {
        float tm = ray.tmax;
        float4 h;
        uint32_t meta;
        if (testRecord<true, KINDS>(s, i, ray, tm, h, meta)) {
            unordered = unordered || h.x != h.x;
            const bool nearer = h.x < tb;
            t2 = nearer ? tb : fminf(t2, h.x);
            if (nearer) { tb = h.x; hit = h; }
        }
    }
    if (COUNT) primsTested += n;
    bool walk = unordered;
    if (__float_as_int(hit.w) >= 0 && !walk) {
        // (b's box is gathered per lane here; carrying it along from the record's uniform load inside the loop was measured slower -- six more
        // v_mov per record some lane hits, in a kernel that is bound by what it issues: 2 392 against 2 498 Msamples/s, profiles/r5_ab_flat_shortcut.txt)
        const float4 blo = s.flat_boxes[2*__float_as_int(hit.w)], bhi = s.flat_boxes[2*__float_as_int(hit.w) + 1];
        const float lo[3] = {blo.x, blo.y, blo.z}, hi[3] = {bhi.x, bhi.y, bhi.z};
        const EmbreeRay e = embreeRay(ray);
        float entry;
        if (embreeLeafEntry(e, lo, hi, entry)) walk = !(tb < t2 && entry <= t2);
        else if (t2 == PT_INF) hit = make_float4(ray.tmax, 0.0f, 0.0f, __int_as_float(-1));
        else walk = true;
    }
    if (walk)
        hit = flatOrderedWalk<KINDS>(s.top_nodes, s.recs, s.objects, ray.o.x, ray.o.y, ray.o.z, ray.tmin, ray.d.x, ray.d.y, ray.d.z, ray.tmax);
    return hit;
}

// `stack` is this lane's column of the workgroup's LDS stack: stack[level*stride]
// FLAT: the scene has <= TGHIP_FLAT_MAX_RECS records; every lane walks the whole record list in step, so the
// record (and quad/cube object) loads have wave-uniform addresses and go through the scalar cache.
template<bool COUNT, bool FLAT, uint32_t KINDS = KINDS_ALL>
PT_DEV float4 traverseClosest(const DeviceScene &s, const RayD &ray, int *stack, int stride,
                              uint32_t &nodesVisited, uint32_t &primsTested, int root = 0)
{
    float tmax = ray.tmax;
    float4 hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
    if (FLAT) {
        if (s.top_nodes)                             // the scene carries the reference's tree: its visiting order decides ties
            return flatClosestOrdered<COUNT, KINDS>(s, ray, primsTested);
        const uint32_t n = s.num_recs;
        for (uint32_t i = 0; i < n; ++i)
            testRecord<true, KINDS>(s, i, ray, tmax, hit);
        if (COUNT) primsTested += n;
        return hit;
    }
    f3 invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
    int sp = 0;
    int cur = root;
    for (;;) {
        if (cur >= 0) {
            const float4 *n = &at32(s.nodes, (uint32_t)cur*4u);
            float4 n0 = ld4(n, 0u), n1 = ld4(n, 1u), n2 = ld4(n, 2u), n3 = ld4(n, 3u);
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
                testRecord<false, KINDS>(s, i, ray, tmax, hit);
            }
        }
        if (sp == 0)
            break;
        sp--;
        cur = stack[sp*stride];
    }
    return hit;
}

// Any-hit query for shadow rays in scenes without forward-lobe BSDFs: true iff some record of an object other
// than `endCap` (the light the ray was aimed at) is hit inside (tmin, tmax).  Equivalent to
// generalizedShadowRay's closest-hit formulation there (TraceBase.cpp:79,114-115), but the
// traversal stops at the first occluder.
template<bool COUNT, bool FLAT, uint32_t KINDS = KINDS_ALL>
PT_DEV bool traverseOccluded(const DeviceScene &s, const RayD &ray, int endCap, int *stack, int stride,
                             uint32_t &nodesVisited, uint32_t &primsTested)
{
    float4 hit;
    uint32_t meta;
    if (FLAT) {
        if (s.top_nodes) {
            // generalizedShadowRay asks for the CLOSEST hit and compares it with the light (TraceBase.cpp:79,114-115): with coincident faces at
            // the light, which record that is follows from the visiting order (flatClosestOrdered), so the any-hit shortcut does not apply
            const float4 h = flatClosestOrdered<COUNT, KINDS>(s, ray, primsTested);
            const int ri = __float_as_int(h.w);
            return ri >= 0 && (int)TGHIP_REC_OBJECT(__float_as_uint(at32(s.recs, (uint32_t)ri*3u).w)) != endCap;
        }
        const uint32_t n = s.num_recs;
        bool occluded = false;
        for (uint32_t i = 0; i < n; ++i) {
            float tmax = ray.tmax;
            if (!occluded) {
                if (COUNT) primsTested++;
                if (testRecord<true, KINDS>(s, i, ray, tmax, hit, meta) && (int)TGHIP_REC_OBJECT(meta) != endCap)
                    occluded = true;
            }
            if (__ballot(!occluded) == 0ull)
                break;                            // every active lane has found its occluder
        }
        return occluded;
    }
    f3 invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
    int sp = 0;
    int cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 *n = &at32(s.nodes, (uint32_t)cur*4u);
            float4 n0 = ld4(n, 0u), n1 = ld4(n, 1u), n2 = ld4(n, 2u), n3 = ld4(n, 3u);
            if (COUNT) nodesVisited++;
            float e0, e1;
            bool h0 = boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ray, invD, ray.tmax, e0);
            bool h1 = boxTest(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), ray, invD, ray.tmax, e1);
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
                float tmax = ray.tmax;
                if (testRecord<false, KINDS>(s, i, ray, tmax, hit, meta) && (int)TGHIP_REC_OBJECT(meta) != endCap)
                    return true;
            }
        }
        if (sp == 0)
            break;
        sp--;
        cur = stack[sp*stride];
    }
    return false;
}

// ---- 8-wide BVH with quantised child boxes (include/tungsten_hip.h: TgHipWideNode) ---------------------------------
// What bounds a per-lane tree walk on MI355X is the number of cache lines it pulls through the CU's vector L1 (one
// 128-byte line every ~2 clocks from L2, tools/ubench_chase.hip; latency hides behind 4 waves per CU already), so
// the walk is built to touch few lines per ray: one 80-byte node decides eight children at once (15.5 -> ~6 node
// visits per ray on materialtest), and a lane asks for exactly ONE thing per loop turn -- a node or a primitive
// record, both behind one base pointer -- so every turn of the wave costs one memory round trip, not one per kind.
// A lane's walk is a pure function of its ray (wideNext / wideVisit below; oracle/oracle.c: wide_walk is the same
// machine), which is what keeps the visit counts of the device and the oracle identical.
//   group  = the hit internal children of the node visited last, in traversal order (bit p = slot p ^ octant), plus
//            that node's imask and first-child index; drained nearest first, the rest waits on the stack
//   triMask = the records of its hit leaf children, tested before the group is touched
struct WideRay { f3 idir; uint32_t octInv; };
PT_DEV WideRay wideRaySetup(const RayD &ray)
{
    WideRay w;
    // no infinities: a plane distance is q*(spacing/d) + (origin - o)/d, and 0*inf would be NaN
    auto inv = [](float d) { return 1.0f/(fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d); };
    w.idir = mk3(inv(ray.d.x), inv(ray.d.y), inv(ray.d.z));
    w.octInv = (w.idir.x < 0.0f ? 1u : 0u) | (w.idir.y < 0.0f ? 2u : 0u) | (w.idir.z < 0.0f ? 4u : 0u);
    return w;
}
PT_DEV uint32_t widePermute(uint32_t h, uint32_t oct)      // bit (s ^ oct) of the result = bit s of h
{
    h = (oct & 1u) ? (((h & 0x55u) << 1) | ((h >> 1) & 0x55u)) : h;
    h = (oct & 2u) ? (((h & 0x33u) << 2) | ((h >> 2) & 0x33u)) : h;
    h = (oct & 4u) ? (((h & 0x0Fu) << 4) | ((h >> 4) & 0x0Fu)) : h;
    return h;
}
struct WideState {
    uint32_t grpBase, grpMasks;   // hits still to visit (bits 0-7, traversal order) | imask << 8
    uint32_t triBase, triMask;    // records still to test: bit positions in the node's leaf_valid
    uint32_t triValid;
    // the DECOUPLED walk of the wavefront kernels (wideNextNode): records of the node visited last while the previous node's are still
    // being tested (tri2Mask != 0: no further node is visited until they have moved up)
    uint32_t tri2Base, tri2Mask, tri2Valid;
    int node;                     // node to visit next; -1: take it from the group / the stack
    int sp;
    // scenes with instance records (the INST variants): the node whose records are being tested, the instance record the walk
    // is inside of (-1: world space)
    uint32_t curNode;
    int curInst;
};
PT_DEV void wideStart(WideState &w) { w.grpBase = 0; w.grpMasks = 0; w.triBase = 0; w.triMask = 0; w.triValid = 0; w.tri2Base = 0; w.tri2Mask = 0; w.tri2Valid = 0; w.node = 0; w.sp = 0; w.curNode = 0; w.curInst = -1; }
// wideStart for the lanes with `go` set, as selects (no region a few lanes enter: a VALU instruction with <= 8 lanes enabled issues at a quarter of
// the rate, profiles/r6_ubench_lane_masks.txt); single-level walks only (curNode / curInst stay)
PT_DEV void wideStartIf(WideState &w, bool go)
{
    w.grpBase = go ? 0u : w.grpBase; w.grpMasks = go ? 0u : w.grpMasks; w.triBase = go ? 0u : w.triBase; w.triMask = go ? 0u : w.triMask;
    w.triValid = go ? 0u : w.triValid; w.tri2Base = go ? 0u : w.tri2Base; w.tri2Mask = go ? 0u : w.tri2Mask; w.tri2Valid = go ? 0u : w.tri2Valid;
    w.node = go ? 0 : w.node; w.sp = go ? 0 : w.sp;
}
// Two-level scenes put two more kinds of entries on the group stack when the walk enters an instance (wideEnterInstance):
#define WIDE_LEAVE     0xFFFFFFFFu   /* x: the master's subtree is done, the ray goes back to world space */
#define WIDE_RECS_FLAG 0x80000000u   /* x = flag | node: records of that (top-level) node still to test, y = their triMask */
// What the lane fetches next: 1 = primitive record `idx`, 2 = node `idx`, 0 = the walk is over; INST also 3 = leave the
// instance (the caller restores the world-space ray), 4 = node `idx`'s record run is needed again (wideResumeRecords).
template<bool INST = false>
PT_DEV int wideNext(WideState &w, uint32_t octInv, uint2 *stack, int stride, uint32_t &idx)
{
    if (w.triMask) {
        const uint32_t b = (uint32_t)__ffs((int)w.triMask) - 1u;
        idx = w.triBase + (uint32_t)__popc(w.triValid & ((1u << b) - 1u));
        w.triMask &= w.triMask - 1u;
        return 1;
    }
    if (w.node < 0) {
        if ((w.grpMasks & 0xFFu) == 0u) {
            if (w.sp == 0)
                return 0;
            w.sp--;
            uint2 e = stack[w.sp*stride];
            if (INST) {
                if (e.x == WIDE_LEAVE) return 3;
                if (e.x & WIDE_RECS_FLAG) { idx = e.x & ~WIDE_RECS_FLAG; w.triMask = e.y; return 4; }
            }
            w.grpBase = e.x; w.grpMasks = e.y;
        }
        uint32_t hits = w.grpMasks & 0xFFu, imask = w.grpMasks >> 8;
        uint32_t slot = ((uint32_t)__ffs((int)hits) - 1u) ^ octInv;
        w.node = (int)(w.grpBase + (uint32_t)__popc(imask & ((1u << slot) - 1u)));
        hits &= hits - 1u;
        w.grpMasks = (imask << 8) | hits;
        if (hits) { stack[w.sp*stride] = make_uint2(w.grpBase, w.grpMasks); w.sp++; }
    }
    idx = (uint32_t)w.node;
    w.node = -1;
    if (INST) w.curNode = idx;
    return 2;
}
// The node half of wideNext alone (single-level scenes): the next node of the walk in `idx`, false when group and stack are empty.
// PT_NEXT_NODE_FLAT = 1 (round 5): the same function with two one-instruction divergent regions (the stack's pop and push, LDS accesses) and selects
// for the rest, instead of three nested branches whose exec-mask bookkeeping and copies were a tenth of the walk's turn.
#ifndef PT_NEXT_NODE_FLAT
#define PT_NEXT_NODE_FLAT 1
#endif
#if PT_NEXT_NODE_FLAT
PT_DEV bool wideNextNode(WideState &w, uint32_t octInv, uint2 *stack, int stride, uint32_t &idx)
{
    const bool need = w.node < 0;                            // the next node comes from the group / the stack
    const bool pop = need && (w.grpMasks & 0xFFu) == 0u && w.sp > 0;
    if (pop) {                                               // (a stacked group always has hits left)
        w.sp--;
        const uint2 e = stack[w.sp*stride];
        w.grpBase = e.x; w.grpMasks = e.y;
    }
    const uint32_t hits = w.grpMasks & 0xFFu, imask = w.grpMasks >> 8;
    const bool have = need && hits != 0u;
    const uint32_t slot = (((uint32_t)__ffs((int)hits) - 1u) ^ octInv) & 7u;
    const uint32_t fromGroup = w.grpBase + (uint32_t)__popc(imask & ((1u << slot) - 1u));
    const uint32_t rest = hits & (hits - 1u);
    w.grpMasks = have ? ((imask << 8) | rest) : w.grpMasks;
    if (have && rest != 0u) { stack[w.sp*stride] = make_uint2(w.grpBase, w.grpMasks); w.sp++; }
    idx = need ? fromGroup : (uint32_t)w.node;
    w.node = -1;
    return !need || have;
}
#else
PT_DEV bool wideNextNode(WideState &w, uint32_t octInv, uint2 *stack, int stride, uint32_t &idx)
{
    if (w.node < 0) {
        if ((w.grpMasks & 0xFFu) == 0u) {
            if (w.sp == 0)
                return false;
            w.sp--;
            uint2 e = stack[w.sp*stride];
            w.grpBase = e.x; w.grpMasks = e.y;
        }
        uint32_t hits = w.grpMasks & 0xFFu, imask = w.grpMasks >> 8;
        uint32_t slot = ((uint32_t)__ffs((int)hits) - 1u) ^ octInv;
        w.node = (int)(w.grpBase + (uint32_t)__popc(imask & ((1u << slot) - 1u)));
        hits &= hits - 1u;
        w.grpMasks = (imask << 8) | hits;
        if (hits) { stack[w.sp*stride] = make_uint2(w.grpBase, w.grpMasks); w.sp++; }
    }
    idx = (uint32_t)w.node;
    w.node = -1;
    return true;
}
#endif
// nothing left to look at: no record, no node, nothing on the stack
PT_DEV bool wideWalkOver(const WideState &w) { return w.triMask == 0u && w.tri2Mask == 0u && w.node < 0 && (w.grpMasks & 0xFFu) == 0u && w.sp == 0; }
// code 4: q1 = the second 16 bytes of node `idx`
PT_DEV void wideResumeRecords(WideState &w, uint32_t idx, float4 q1)
{
    w.triBase = __float_as_uint(q1.y);
    w.triValid = __float_as_uint(q1.z);
    w.curNode = idx;
}
// The record just fetched (r0, r1, r2) is an instance: what is left of the current node goes on the stack, the ray into the
// master's space (rotation + translation: distances along it are unchanged; primitives/Instance.cpp:290-311), the walk on to
// the master's wide subtree.
PT_DEV void wideEnterInstance(WideState &w, uint2 *stack, int stride, uint32_t recIdx, float4 r0, float4 r1, float4 r2, RayD &ray, WideRay &wr)
{
    if (w.grpMasks & 0xFFu) { stack[w.sp*stride] = make_uint2(w.grpBase, w.grpMasks); w.sp++; }
    if (w.triMask) { stack[w.sp*stride] = make_uint2(WIDE_RECS_FLAG | w.curNode, w.triMask); w.sp++; }
    stack[w.sp*stride] = make_uint2(WIDE_LEAVE, 0u); w.sp++;
    w.grpMasks = 0u; w.triMask = 0u;
    const f3 qc = -xyz(r1);                             // conjugate(): the inverse rotation
    ray.o = quatRotate(r1.w, qc, ray.o - xyz(r0));
    ray.d = quatRotate(r1.w, qc, ray.d);
    wr = wideRaySetup(ray);
    w.node = (int)__float_as_uint(r2.z);
    w.curInst = (int)recIdx;
}
// ---- the node as a lane holds it --------------------------------------------------------------------------------------------------
// PT_WIDE_HALF = 1 (an experiment of round 5, measured and NOT the product's layout): the shim re-encodes the ABI's 80-byte node
// (include/tungsten_hip.h: TgHipWideNode) at upload into 128 bytes = one cache line whose 48 child planes are IEEE halfs holding the SAME
// integers 0 .. 255 -- exactly --, one 16-byte row per axis and side:
//   [0,16) origin.xyz, exp | imask << 24    [16,32) child_base, rec_base, leaf_valid, 0
//   [32,48) lo x   [48,64) lo y   [64,80) lo z   [80,96) hi x   [96,112) hi y   [112,128) hi z        (8 halfs per row: slots 0 .. 7)
// The idea: a wave64 v_cvt_f32_ubyteN costs 3.2 cycles of its SIMD and the v_pk_fma_f32 behind it 3.7 per two planes (tools/ubench_valu.hip,
// profiles/r5_ubench_valu.txt: only fma / mul / add / and / mov run at 2) -- 5.0 per plane --, while v_fma_mix_f32 reads the half and does
// the same single-rounding f32 fma in 3.2; and the ray's octant picks the ROW it loads as "near" / "far" per axis by address, where the byte
// layout needs twelve v_cndmask_b32 on the loaded words: ~85 of ~500 issue cycles per node visit less, every plane distance bit-identical
// (the whole GPU suite passes on it).  The result: 2.7 % SLOWER on the headline (949 against 975 Msamples/s in one session, closest-hit launches
// 863 -> 890 us, shadow 827 -> 888; mesh1m 610 -> 600; profiles/r5_ab_half_planes.txt) -- eight 16-byte loads per node instead of five, and
// 128 VGPRs instead of 115 / 125: the walk is bound as much by what it pulls through the vector L1 as by what it issues.
#ifndef PT_WIDE_HALF
#define PT_WIDE_HALF 0
#endif
// byte offset of node `idx` behind s.wide
PT_DEV uint32_t wideNodeOff(const DeviceScene &s, uint32_t idx) { return PT_WIDE_HALF ? idx << 7 : idx*s.wide_stride; }
#if PT_WIDE_HALF
#define PT_WIDE_NODE_BYTES 128u
struct WideNodeRegs { float4 q0, q1, nx, ny, nz, fx, fy, fz; };
// the 16-byte row at byte `off` behind `base` (uniform: s.wide, or the LDS copy of the top of the tree) -- a 32-bit offset, so that the
// load is `global_load_dwordx4 v, v_off, s[base]` and a lane holds six row offsets, not six 64-bit pointers
PT_DEV float4 wideRow(const char *base, uint32_t off) { return *reinterpret_cast<const float4 *>(base + (size_t)off); }
PT_DEV void wideNodeFetch(WideNodeRegs &n, const char *base, uint32_t off, const WideRay &wr)
{
    const uint32_t ox = (wr.octInv & 1u) ? 48u : 0u, oy = (wr.octInv & 2u) ? 48u : 0u, oz = (wr.octInv & 4u) ? 48u : 0u;
    n.q0 = wideRow(base, off); n.q1 = wideRow(base, off + 16u);
    n.nx = wideRow(base, off + 32u + ox); n.fx = wideRow(base, off + 80u - ox);
    n.ny = wideRow(base, off + 48u + oy); n.fy = wideRow(base, off + 96u - oy);
    n.nz = wideRow(base, off + 64u + oz); n.fz = wideRow(base, off + 112u - oz);
}
// (the two-level walks load "a node or a record" through one address: the first three rows are in registers when the kind is known)
PT_DEV void wideNodeFetchRest(WideNodeRegs &n, const char *base, uint32_t off, const WideRay &wr, float4 q0, float4 q1, float4 q2)
{
    const uint32_t ox = (wr.octInv & 1u) ? 48u : 0u, oy = (wr.octInv & 2u) ? 48u : 0u, oz = (wr.octInv & 4u) ? 48u : 0u;
    const float4 hx = wideRow(base, off + 80u);
    n.q0 = q0; n.q1 = q1;
    n.nx = ox ? hx : q2; n.fx = ox ? q2 : hx;
    n.ny = wideRow(base, off + 48u + oy); n.fy = wideRow(base, off + 96u - oy);
    n.nz = wideRow(base, off + 64u + oz); n.fz = wideRow(base, off + 112u - oz);
}
#else
#define PT_WIDE_NODE_BYTES 80u
struct WideNodeRegs { float4 q0, q1, q2, q3, q4; };
// (a 16-byte row is loaded as ONE vector: read through float4 the compiler narrows the row whose last word the walk does not use -- child base,
// record base, leaf_valid, - -- to a global_load_dwordx3, and the walk built around the 12-byte load is the slower one (pt_math.h: ld4): PT_ROW_X4 = 0 for the A/B)
#ifndef PT_ROW_X4
#define PT_ROW_X4 1
#endif
PT_DEV float4 wideRow(const char *base, uint32_t off)
{
#if PT_ROW_X4
    const PtF4v v = *reinterpret_cast<const PtF4v *>(base + (size_t)off);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4 *>(base + (size_t)off);
#endif
}
PT_DEV void wideNodeFetch(WideNodeRegs &n, const char *base, uint32_t off, const WideRay &)
{
    n.q0 = wideRow(base, off); n.q1 = wideRow(base, off + 16u); n.q2 = wideRow(base, off + 32u); n.q3 = wideRow(base, off + 48u); n.q4 = wideRow(base, off + 64u);
}
PT_DEV void wideNodeFetchRest(WideNodeRegs &n, const char *base, uint32_t off, const WideRay &, float4 q0, float4 q1, float4 q2)
{
    n.q0 = q0; n.q1 = q1; n.q2 = q2; n.q3 = wideRow(base, off + 48u); n.q4 = wideRow(base, off + 64u);
}
#endif
// The node's 80 bytes have arrived: slab-test its eight children against [tmin, tmax], queue the hit ones.  Two children per
// v_pk_fma_f32; each half is one correctly rounded fma, as in the oracle's scalar fmaf.
typedef float WideF2 __attribute__((ext_vector_type(2)));
// skip (wave-uniform): records named in the node's `reserved` word are not queued (the scene's hoisted quad, DeviceScene::hoisted_rec)
PT_DEV void wideVisit(WideState &w, const WideNodeRegs &nd, f3 o, const WideRay &wr, float tmin, float tmax, bool skip = false)
{
    const float4 q0 = nd.q0, q1 = nd.q1;
    const uint32_t ex = __float_as_uint(q0.w);
    const f3 spacing = mk3(__uint_as_float((ex & 0xFFu) << 23), __uint_as_float(((ex >> 8) & 0xFFu) << 23), __uint_as_float(((ex >> 16) & 0xFFu) << 23));
    const f3 adjS = spacing*wr.idir;
    const f3 adjO = (xyz(q0) - o)*wr.idir;
    uint32_t hitmask = 0u;
#if PT_WIDE_HALF
    typedef _Float16 WideH2 __attribute__((ext_vector_type(2)));
    // plane distance = fma(q, spacing/d, (origin - o)/d) with q read as a half (v_fma_mix_f32): one rounding, as the oracle's fmaf on float(q)
    auto plane = [](_Float16 q, float s, float o) { return __builtin_fmaf((float)q, s, o); };
    const float nxw[4] = {nd.nx.x, nd.nx.y, nd.nx.z, nd.nx.w}, nyw[4] = {nd.ny.x, nd.ny.y, nd.ny.z, nd.ny.w}, nzw[4] = {nd.nz.x, nd.nz.y, nd.nz.z, nd.nz.w};
    const float fxw[4] = {nd.fx.x, nd.fx.y, nd.fx.z, nd.fx.w}, fyw[4] = {nd.fy.x, nd.fy.y, nd.fy.z, nd.fy.w}, fzw[4] = {nd.fz.x, nd.fz.y, nd.fz.z, nd.fz.w};
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {                // children 2 pr and 2 pr + 1: the two halfs of word pr of every row
        const WideH2 hnx = __builtin_bit_cast(WideH2, nxw[pr]), hny = __builtin_bit_cast(WideH2, nyw[pr]), hnz = __builtin_bit_cast(WideH2, nzw[pr]);
        const WideH2 hfx = __builtin_bit_cast(WideH2, fxw[pr]), hfy = __builtin_bit_cast(WideH2, fyw[pr]), hfz = __builtin_bit_cast(WideH2, fzw[pr]);
        const float tnx0 = plane(hnx.x, adjS.x, adjO.x), tnx1 = plane(hnx.y, adjS.x, adjO.x), tfx0 = plane(hfx.x, adjS.x, adjO.x), tfx1 = plane(hfx.y, adjS.x, adjO.x);
        const float tny0 = plane(hny.x, adjS.y, adjO.y), tny1 = plane(hny.y, adjS.y, adjO.y), tfy0 = plane(hfy.x, adjS.y, adjO.y), tfy1 = plane(hfy.y, adjS.y, adjO.y);
        const float tnz0 = plane(hnz.x, adjS.z, adjO.z), tnz1 = plane(hnz.y, adjS.z, adjO.z), tfz0 = plane(hfz.x, adjS.z, adjO.z), tfz1 = plane(hfz.y, adjS.z, adjO.z);
        float tn0 = fmaxf(fmaxf(tnx0, tny0), fmaxf(tnz0, tmin)), tn1 = fmaxf(fmaxf(tnx1, tny1), fmaxf(tnz1, tmin));
        float tf0 = fminf(fminf(tfx0, tfy0), fminf(tfz0, tmax)), tf1 = fminf(fminf(tfx1, tfy1), fminf(tfz1, tmax));
        tf0 *= 1.0000004f; tf1 *= 1.0000004f;
        hitmask |= (tn0 <= tf0) ? (1u << (2*pr)) : 0u;
        hitmask |= (tn1 <= tf1) ? (2u << (2*pr)) : 0u;
    }
#else
    const float4 q2 = nd.q2, q3 = nd.q3, q4 = nd.q4;
    const WideF2 SX = {adjS.x, adjS.x}, SY = {adjS.y, adjS.y}, SZ = {adjS.z, adjS.z};
    const WideF2 OX = {adjO.x, adjO.x}, OY = {adjO.y, adjO.y}, OZ = {adjO.z, adjO.z};
    // the planes the ray enters (near) and leaves (far) through, per axis: qlo / qhi swapped for negative directions
    const bool nx = (wr.octInv & 1u) != 0u, ny = (wr.octInv & 2u) != 0u, nz = (wr.octInv & 4u) != 0u;
    const uint32_t lx0 = __float_as_uint(q2.x), lx1 = __float_as_uint(q2.y), ly0 = __float_as_uint(q2.z), ly1 = __float_as_uint(q2.w);
    const uint32_t lz0 = __float_as_uint(q3.x), lz1 = __float_as_uint(q3.y), hx0 = __float_as_uint(q3.z), hx1 = __float_as_uint(q3.w);
    const uint32_t hy0 = __float_as_uint(q4.x), hy1 = __float_as_uint(q4.y), hz0 = __float_as_uint(q4.z), hz1 = __float_as_uint(q4.w);
    const uint32_t nearX[2] = {nx ? hx0 : lx0, nx ? hx1 : lx1}, farX[2] = {nx ? lx0 : hx0, nx ? lx1 : hx1};
    const uint32_t nearY[2] = {ny ? hy0 : ly0, ny ? hy1 : ly1}, farY[2] = {ny ? ly0 : hy0, ny ? ly1 : hy1};
    const uint32_t nearZ[2] = {nz ? hz0 : lz0, nz ? hz1 : lz1}, farZ[2] = {nz ? lz0 : hz0, nz ? lz1 : hz1};
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {                // children 2 pr and 2 pr + 1
        const int d = pr >> 1, sh = (pr & 1)*16;
        auto pair = [&](uint32_t word) { WideF2 v = {(float)((word >> sh) & 0xFFu), (float)((word >> (sh + 8)) & 0xFFu)}; return v; };
        const WideF2 tnx = __builtin_elementwise_fma(pair(nearX[d]), SX, OX), tfx = __builtin_elementwise_fma(pair(farX[d]), SX, OX);
        const WideF2 tny = __builtin_elementwise_fma(pair(nearY[d]), SY, OY), tfy = __builtin_elementwise_fma(pair(farY[d]), SY, OY);
        const WideF2 tnz = __builtin_elementwise_fma(pair(nearZ[d]), SZ, OZ), tfz = __builtin_elementwise_fma(pair(farZ[d]), SZ, OZ);
        float tn0 = fmaxf(fmaxf(tnx.x, tny.x), fmaxf(tnz.x, tmin)), tn1 = fmaxf(fmaxf(tnx.y, tny.y), fmaxf(tnz.y, tmin));
        float tf0 = fminf(fminf(tfx.x, tfy.x), fminf(tfz.x, tmax)), tf1 = fminf(fminf(tfx.y, tfy.y), fminf(tfz.y, tmax));
        tf0 *= 1.0000004f; tf1 *= 1.0000004f;
        hitmask |= (tn0 <= tf0) ? (1u << (2*pr)) : 0u;
        hitmask |= (tn1 <= tf1) ? (2u << (2*pr)) : 0u;
    }
#endif
    const uint32_t imask = ex >> 24;
    // the records of the hit leaf children: bit s -> bits 4 s .. 4 s + 3, masked by the records that exist
    uint32_t x = hitmask & ~imask;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    x = (x << 4) - x;
    w.grpBase = __float_as_uint(q1.x);
    w.grpMasks = (imask << 8) | widePermute(hitmask & imask, wr.octInv);
    w.triBase = __float_as_uint(q1.y);
    w.triValid = __float_as_uint(q1.z);
    w.triMask = x & w.triValid;
    w.triMask &= ~(skip ? __float_as_uint(q1.w) : 0u);
}
// ---- suspended walks (PathState::suspend_*) -------------------------------------------------------------------------
// The flag that a queued ray / shadow slot is a suspended walk is the sign bit of its tmin word (A_RAY_O.w / A_SH_O.w: 1e-4, 5e-4 or 0).
#define WALK_SUSPENDED_BIT 0x80000000u
PT_DEV void walkSave(const PathState &st, uint32_t slot, const WideState &w, const uint2 *stack, int stride)
{
    slotU4(st, st.walk_base, slot) = make_uint4(w.grpBase, w.grpMasks, w.triBase, w.triMask);
    slotU4(st, st.walk_base + 1u, slot) = make_uint4(w.triValid, (uint32_t)w.node, (uint32_t)w.sp, 0u);
    slotU4(st, st.walk_base + 3u, slot) = make_uint4(w.tri2Base, w.tri2Mask, w.tri2Valid, 0u);
    for (int l = 0; l < w.sp; l += 2) {
        const uint2 a = stack[l*stride], b = l + 1 < w.sp ? stack[(l + 1)*stride] : make_uint2(0u, 0u);
        slotU4(st, st.walk_base + 4u + (uint32_t)(l >> 1), slot) = make_uint4(a.x, a.y, b.x, b.y);
    }
}
PT_DEV void walkRestore(const PathState &st, uint32_t slot, WideState &w, uint2 *stack, int stride)
{
    const uint4 a = slotU4(st, st.walk_base, slot), b = slotU4(st, st.walk_base + 1u, slot);
    w.grpBase = a.x; w.grpMasks = a.y; w.triBase = a.z; w.triMask = a.w;
    w.triValid = b.x; w.node = (int)b.y; w.sp = (int)b.z;
    const uint4 c = slotU4(st, st.walk_base + 3u, slot);
    w.tri2Base = c.x; w.tri2Mask = c.y; w.tri2Valid = c.z;
    w.curNode = 0; w.curInst = -1;
    for (int l = 0; l < w.sp; l += 2) {
        const uint4 e = slotU4(st, st.walk_base + 4u + (uint32_t)(l >> 1), slot);
        stack[l*stride] = make_uint2(e.x, e.y);
        if (l + 1 < w.sp) stack[(l + 1)*stride] = make_uint2(e.z, e.w);
    }
}
PT_DEV const float4 *wideNodePtr(const DeviceScene &s, uint32_t idx) { return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.wide) + (size_t)(wideNodeOff(s, idx))); }

// closest hit through the wide BVH, one ray at a time (tghip_trace_rays); the dynamic-fetch kernels run the same machine
template<bool COUNT, uint32_t KINDS = KINDS_ALL, bool INST = false>
PT_DEV float4 traverseClosestWide(const DeviceScene &s, const RayD &worldRay, uint2 *stack, int stride, uint32_t &nodesVisited, uint32_t &primsTested, int &hitInst)
{
    float tmax = worldRay.tmax;
    float4 hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
    RayD ray = worldRay;
    WideRay wr = wideRaySetup(ray);
    WideState w;
    wideStart(w);
    hitInst = -1;
    for (;;) {
        uint32_t idx;
        const int what = wideNext<INST>(w, wr.octInv, stack, stride, idx);
        if (what == 0)
            break;
        if (what == 2) {
            WideNodeRegs nd;
            wideNodeFetch(nd, reinterpret_cast<const char *>(s.wide), wideNodeOff(s, idx), wr);
            if (COUNT) nodesVisited++;
            wideVisit(w, nd, ray.o, wr, ray.tmin, tmax);
        } else if (what == 1) {
            if (COUNT) primsTested++;
            float4 r0 = ld4(s.recs, idx*3u + 0u), r1 = ld4(s.recs, idx*3u + 1u), r2 = ld4(s.recs, idx*3u + 2u);
            if (INST && TGHIP_REC_KIND(__float_as_uint(r0.w)) == TGHIP_REC_INSTANCE) {
                wideEnterInstance(w, stack, stride, idx, r0, r1, r2, ray, wr);
            } else {
                uint32_t meta;
                if (testRecordLoaded<false, KINDS>(s, idx, r0, r1, r2, ray, tmax, hit, meta))
                    hitInst = w.curInst;
            }
        } else if (what == 3) {
            ray.o = worldRay.o; ray.d = worldRay.d;
            wr = wideRaySetup(ray);
            w.curInst = -1;
        } else {
            wideResumeRecords(w, idx, wideNodePtr(s, idx)[1]);
        }
    }
    return hit;
}

// ---- scenes with `instances` primitives: Instance::intersect as the reference computes it (primitives/Instance.cpp:290-311) --------
// The scene's BVH2 (nodes[0]) holds the non-instance records and one TGHIP_REC_INSTANCE_SET record per `instances` primitive.  Reaching
// one runs BinaryBvh::trace (bvh/BinaryBvh.hpp:197-287) over the reference's OWN tree over the instances -- restated node for node by the
// host (csrc/host/RefInstanceBvh.cpp), stored as BVH2 nodes with its exact child boxes, leaf references into inst_prims --: every
// instance whose leaf the walk reaches gets the ray in its master's space with nearT = the distance at which the ray enters that leaf's
// box and farT = INFINITY (Ray::scatter's default), and a hit there REPLACES the hit so far, nearer or not; the walk's own tMax is the
// minimum of the hit distances so far and culls the boxes that begin behind it.  So the result is the last hit in that tree's visiting
// order -- the tree, the near / far rule (`minMax[0] < minMax[1]`: the right child first on a tie), the entry distance kept with every
// stacked node and re-checked when it is popped are all the reference's (oracle/oracle.c: instance_set_walk is the same statements).
PT_DEV float refMax(float a, float b) { return a > b ? a : b; }      // _mm_max_ps / Tungsten::max: the SECOND operand when unordered
PT_DEV float refMin(float a, float b) { return a < b ? a : b; }
// BinaryBvh::bboxIntersection (:155-178)
PT_DEV bool refBboxIntersection(f3 lo, f3 hi, const RayD &ray, float &tMin, float &tMax)
{
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z};
    const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    float ttMin = tMin, ttMax = tMax;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float invD = 1.0f/d[i], relMin = l[i] - o[i], relMax = h[i] - o[i];
        if (invD >= 0.0f) {
            ttMin = refMax(ttMin, relMin*invD);
            ttMax = refMin(ttMax, relMax*invD);
        } else {
            ttMax = refMin(ttMax, relMin*invD);
            ttMin = refMax(ttMin, relMax*invD);
        }
    }
    if (ttMin <= ttMax) { tMin = ttMin; tMax = ttMax; return true; }
    return false;
}
// one child of a node in BinaryBvh::trace (:231-242): the entry distance and, negated as the reference carries it, the exit distance,
// clipped to [nearT, farT]; near / far planes by the sign of the direction (>= 0: keep), products with 1/d and -1/d
PT_DEV bool refChildTest(f3 lo, f3 hi, f3 o, f3 d, f3 invD, float nearT, float farT, float &tEntry)
{
    const float tnx = ((d.x >= 0.0f ? lo.x : hi.x) - o.x)*invD.x, ntfx = ((d.x >= 0.0f ? hi.x : lo.x) - o.x)*(-invD.x);
    const float tny = ((d.y >= 0.0f ? lo.y : hi.y) - o.y)*invD.y, ntfy = ((d.y >= 0.0f ? hi.y : lo.y) - o.y)*(-invD.y);
    const float tnz = ((d.z >= 0.0f ? lo.z : hi.z) - o.z)*invD.z, ntfz = ((d.z >= 0.0f ? hi.z : lo.z) - o.z)*(-invD.z);
    const float tmin = refMax(refMax(refMax(tnx, tny), tnz), nearT);
    const float ntmax = refMax(refMax(refMax(ntfx, ntfy), ntfz), -farT);
    tEntry = tmin;
    return tmin <= -ntmax;
}
// the ray in an instance's master space (Instance.cpp:295-296: rotation and translation only, so distances along it are unchanged)
PT_DEV void instanceLocalRay(const DeviceScene &s, uint32_t ri, const RayD &world, float tmin, float tmax, RayD &local, int &masterRoot)
{
    float4 r0 = ld4(s.recs, ri*3u + 0u), r1 = ld4(s.recs, ri*3u + 1u), r2 = ld4(s.recs, ri*3u + 2u);
    f3 qc = -xyz(r1);                                   // conjugate(): the inverse rotation
    local.o = quatRotate(r1.w, qc, world.o - xyz(r0));
    local.d = quatRotate(r1.w, qc, world.d);
    local.tmin = tmin; local.tmax = tmax;
    masterRoot = (int)__float_as_uint(r2.x);
}
// The distance at which the ray enters the leaf at inst_prims slot `first`, as its parent's test computed it (same box, same arithmetic:
// the entry distance does not depend on farT).  BinaryBvh::trace keeps it with every stacked node; here it is recomputed when a stacked
// LEAF is popped -- inner nodes need none: the children of a node that begins behind tMax all begin behind it too (float subtraction and
// multiplication are monotonic and the builder's boxes nest), i.e. fail their own tests exactly when the reference's pop test would have
// dropped the node -- so the stack holds one word per level.
PT_DEV float refLeafEntry(const DeviceScene &s, uint32_t first, f3 o, f3 d, f3 invD, float nearT)
{
    const float4 blo = s.inst_leaf_boxes[2u*first], bhi = s.inst_leaf_boxes[2u*first + 1u];
    float tEntry;
    (void)refChildTest(xyz(blo), xyz(bhi), o, d, invD, nearT, PT_INF, tEntry);
    return tEntry;
}
// Can the ray hit instance record `ri` at all, anywhere beyond `tmin`?  Its geometry lies inside inst_tight_boxes[ri] (padded for the
// rounding of the world -> master transform, csrc/host/Scene.cpp: tightenInstanceBounds), so a ray that misses that box need not be
// taken into the master: the reference takes it there and finds nothing.  An optimisation only -- two thirds of the instances the
// reference's tree lets a ray into are entered in vain, their leaf's box being the union of the rotated CORNERS of up to two master boxes.
PT_DEV bool instanceReachable(const DeviceScene &s, uint32_t ri, const RayD &world, f3 winvD, float tmin)
{
    const float4 blo = s.inst_tight_boxes[2u*ri], bhi = s.inst_tight_boxes[2u*ri + 1u];
    RayD r = world;
    r.tmin = tmin;
    float e;
    return boxTest(xyz(blo), xyz(bhi), r, winvD, PT_INF, e);
}
// Instance::intersect for the set record `setRec`: `tmax` is the ray's farT (it may GROW here), `hit` / `hitInst` the hit so far.
// `stack` is the part of this lane's stack above what the caller has pushed; the master's walk sits above the instance tree's.
template<bool COUNT, uint32_t KINDS>
PT_DEV void instanceSetIntersect(const DeviceScene &s, uint32_t setRec, const RayD &ray, float &tmax, float4 &hit, int &hitInst,
                                 int *stack, int stride, uint32_t &nodesVisited, uint32_t &primsTested)
{
    const float4 s0 = ld4(s.recs, setRec*3u + 0u), s1 = ld4(s.recs, setRec*3u + 1u), s2 = ld4(s.recs, setRec*3u + 2u);
    if (COUNT) primsTested++;
    float tMin = ray.tmin, tMax = tmax;
    if (!refBboxIntersection(xyz(s0), xyz(s1), ray, tMin, tMax))
        return;
    const f3 invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
    float farT = tmax;                                  // nearFar[2..3], negated: the ray's farT until the first leaf, the walk's tMax after it
    int node = __float_as_int(s2.x);
    int sp = 0;
    for (;;) {
        bool miss = false;
        while (node >= 0) {
            const float4 *n = &at32(s.nodes, (uint32_t)node*4u);
            const float4 n0 = ld4(n, 0u), n1 = ld4(n, 1u), n2 = ld4(n, 2u), n3 = ld4(n, 3u);
            if (COUNT) nodesVisited++;
            float e0, e1;
            const bool hitL = refChildTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ray.o, ray.d, invD, ray.tmin, farT, e0);
            const bool hitR = refChildTest(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), ray.o, ray.d, invD, ray.tmin, farT, e1);
            const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            if (hitL && hitR) {
                if (e0 < e1) { stack[sp*stride] = c1; ++sp; node = c0; tMin = e0; }
                else         { stack[sp*stride] = c0; ++sp; node = c1; tMin = e1; }
            } else if (hitL) { node = c0; tMin = e0; }
            else if (hitR) { node = c1; tMin = e1; }
            else { miss = true; break; }
        }
        if (!miss) {
            const uint32_t first = TGHIP_LEAF_FIRST(node), count = TGHIP_LEAF_COUNT(node);
            for (uint32_t k = first; k < first + count; ++k) {
                const uint32_t ri = s.inst_prims[k];
                if (COUNT) primsTested++;
                if (!instanceReachable(s, ri, ray, invD, tMin))
                    continue;
                RayD local;
                int root;
                instanceLocalRay(s, ri, ray, tMin, PT_INF, local, root);
                const float4 lh = traverseClosest<COUNT, false, KINDS>(s, local, stack + sp*stride, stride, nodesVisited, primsTested, root);
                if (__float_as_int(lh.w) >= 0) { hit = lh; hitInst = (int)ri; tmax = lh.x; }    // ray.setFarT(localRay.farT())
            }
            tMax = refMin(tMax, tmax);
            farT = tMax;
        }
        for (;;) {                                       // pop: a leaf that begins behind tMax is dropped (inner nodes: see refLeafEntry)
            if (sp == 0) return;
            --sp;
            node = stack[sp*stride];
            if (node >= 0) break;
            tMin = refLeafEntry(s, TGHIP_LEAF_FIRST(node), ray.o, ray.d, invD, ray.tmin);
            if (!(tMax < tMin)) break;
        }
    }
}

template<bool COUNT, uint32_t KINDS = KINDS_ALL>
PT_DEV float4 traverseClosestInst(const DeviceScene &s, const RayD &ray, int *stack, int stride,
                                  uint32_t &nodesVisited, uint32_t &primsTested, int &hitInst)
{
    float tmax = ray.tmax;
    float4 hit = make_float4(tmax, 0.0f, 0.0f, __int_as_float(-1));
    hitInst = -1;
    const f3 invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
    int sp = 0, cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 *n = &at32(s.nodes, (uint32_t)cur*4u);
            float4 n0 = ld4(n, 0u), n1 = ld4(n, 1u), n2 = ld4(n, 2u), n3 = ld4(n, 3u);
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
                if (TGHIP_REC_KIND(__float_as_uint(at32(s.recs, i*3u).w)) == TGHIP_REC_INSTANCE_SET) {
                    instanceSetIntersect<COUNT, KINDS>(s, i, ray, tmax, hit, hitInst, stack + sp*stride, stride, nodesVisited, primsTested);
                    continue;
                }
                if (COUNT) primsTested++;
                uint32_t meta;
                if (testRecord<false, KINDS>(s, i, ray, tmax, hit, meta))
                    hitInst = -1;
            }
        }
        if (sp == 0)
            break;
        sp--;
        cur = stack[sp*stride];
    }
    return hit;
}

// Any-hit query through the BVH2 of a scene with `instances` primitives.  Instance::intersect answers "hit" as soon as ONE instance whose
// leaf the walk reaches is hit anywhere beyond that leaf's entry distance -- its master gets farT = infinity, so geometry BEHIND the
// light occludes it as long as its leaf's box begins in front (a property of the reference this keeps) -- and before the first hit the
// walk's farT does not move, so which leaves are reached does not depend on the visiting order: every leaf whose box passes the
// reference's test against [nearT, farT].
template<bool COUNT, uint32_t KINDS = KINDS_ALL>
PT_DEV bool instanceSetOccluded(const DeviceScene &s, uint32_t setRec, const RayD &ray, int *stack, int stride, uint32_t &nodesVisited, uint32_t &primsTested)
{
    const float4 s0 = ld4(s.recs, setRec*3u + 0u), s1 = ld4(s.recs, setRec*3u + 1u), s2 = ld4(s.recs, setRec*3u + 2u);
    if (COUNT) primsTested++;
    float tMin = ray.tmin, tMax = ray.tmax;
    if (!refBboxIntersection(xyz(s0), xyz(s1), ray, tMin, tMax))
        return false;
    const f3 invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
    int node = __float_as_int(s2.x);
    int sp = 0;
    for (;;) {
        bool miss = false;
        while (node >= 0) {
            const float4 *n = &at32(s.nodes, (uint32_t)node*4u);
            const float4 n0 = ld4(n, 0u), n1 = ld4(n, 1u), n2 = ld4(n, 2u), n3 = ld4(n, 3u);
            if (COUNT) nodesVisited++;
            float e0, e1;
            const bool hitL = refChildTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ray.o, ray.d, invD, ray.tmin, ray.tmax, e0);
            const bool hitR = refChildTest(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), ray.o, ray.d, invD, ray.tmin, ray.tmax, e1);
            const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            if (hitL && hitR) { stack[sp*stride] = c1; ++sp; node = c0; tMin = e0; }
            else if (hitL) { node = c0; tMin = e0; }
            else if (hitR) { node = c1; tMin = e1; }
            else { miss = true; break; }
        }
        if (!miss) {
            const uint32_t first = TGHIP_LEAF_FIRST(node), count = TGHIP_LEAF_COUNT(node);
            for (uint32_t k = first; k < first + count; ++k) {
                const uint32_t ri = s.inst_prims[k];
                if (COUNT) primsTested++;
                if (!instanceReachable(s, ri, ray, invD, tMin))
                    continue;
                RayD local;
                int root;
                instanceLocalRay(s, ri, ray, tMin, PT_INF, local, root);
                const float4 lh = traverseClosest<COUNT, false, KINDS>(s, local, stack + sp*stride, stride, nodesVisited, primsTested, root);
                if (__float_as_int(lh.w) >= 0)
                    return true;
            }
        }
        if (sp == 0) return false;
        --sp;
        node = stack[sp*stride];
        if (node < 0) tMin = refLeafEntry(s, TGHIP_LEAF_FIRST(node), ray.o, ray.d, invD, ray.tmin);
    }
}
template<bool COUNT, uint32_t KINDS = KINDS_ALL>
PT_DEV bool traverseOccludedInst(const DeviceScene &s, const RayD &ray, int endCap, int *stack, int stride,
                                 uint32_t &nodesVisited, uint32_t &primsTested)
{
    const f3 invD = mk3(1.0f/ray.d.x, 1.0f/ray.d.y, 1.0f/ray.d.z);
    float4 hit;
    int sp = 0, cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 *n = &at32(s.nodes, (uint32_t)cur*4u);
            float4 n0 = ld4(n, 0u), n1 = ld4(n, 1u), n2 = ld4(n, 2u), n3 = ld4(n, 3u);
            if (COUNT) nodesVisited++;
            float e0, e1;
            bool h0 = boxTest(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), ray, invD, ray.tmax, e0);
            bool h1 = boxTest(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), ray, invD, ray.tmax, e1);
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
                if (TGHIP_REC_KIND(__float_as_uint(at32(s.recs, i*3u).w)) == TGHIP_REC_INSTANCE_SET) {
                    if (instanceSetOccluded<COUNT, KINDS>(s, i, ray, stack + sp*stride, stride, nodesVisited, primsTested))
                        return true;
                    continue;
                }
                if (COUNT) primsTested++;
                uint32_t meta;
                float tmax = ray.tmax;
                if (testRecord<false, KINDS>(s, i, ray, tmax, hit, meta) && (int)TGHIP_REC_OBJECT(meta) != endCap)
                    return true;
            }
        }
        if (sp == 0)
            break;
        sp--;
        cur = stack[sp*stride];
    }
    return false;
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
