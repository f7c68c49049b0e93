// path_tracer_hip: kernels + the extern "C" shim declared in include/tungsten_hip.h.
// gfx950 (MI355X) only.  See pt_kernels.h for the execution model and DESIGN.md for the layout.
#define PT_WAVEFRONT_MAIN
#include "pt_wavefront.h"

#include <atomic>
#include <chrono>
#include <map>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is loaded with dlopen at the first multi-GPU reduce

// The k_shade variants are instantiated in shade_simple.hip / shade_class.hip / shade_full.hip (parallel compilation).
extern template __global__ void k_shade<MASK_LEAN, LEAN_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_LEAN | FEAT_QMC), LEAN_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_LEAN, LEAN_WAVES, 3>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_LEAN | FEAT_QMC), LEAN_WAVES, 3>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_LEAN, LEAN_WAVES, 7>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_LEAN | FEAT_QMC), LEAN_WAVES, 7>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_SIMPLE, SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_SIMPLE | FEAT_QMC), SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_SIMPLE_INST, SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_SIMPLE_INST | FEAT_QMC), SIMPLE_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_SIMPLE, SIMPLE_WAVES, 3>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_SIMPLE | FEAT_QMC), SIMPLE_WAVES, 3>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_SIMPLE, SIMPLE_WAVES, 7>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_SIMPLE | FEAT_QMC), SIMPLE_WAVES, 7>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_COAT, COAT_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_COAT | FEAT_QMC), COAT_WAVES, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_COAT, COAT_WAVES, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_COAT | FEAT_QMC), COAT_WAVES, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_GLASS, 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_GLASS | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_GLASS, 2, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_GLASS | FEAT_QMC), 2, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_PLASTIC, 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_PLASTIC | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_PLASTIC, 2, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_PLASTIC | FEAT_QMC), 2, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_COAT_INST, 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_COAT_INST | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_GLASS_INST, 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_GLASS_INST | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_PLASTIC_INST, 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_PLASTIC_INST | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_FULL, 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_FULL | FEAT_QMC), 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_FULL, 2, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<(MASK_FULL | FEAT_QMC), 2, 2>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<BSDF_MASK_ALL, 2, 0>(DeviceScene, PathState, PassParams, int);
extern template __global__ void k_shade<MASK_MEDIA, 2, 0>(DeviceScene, PathState, PassParams, int);   // shade_media.hip
extern template __global__ void k_shade<BSDF_MASK_ALL, 2, 0, true>(DeviceScene, PathState, PassParams, int);   // shade_global.hip
// k_tail: tail.hip
// walk_shadow.hip (compiled without SLP vectorisation)
extern template __global__ void k_trace_shadow_fast<false, false>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_trace_shadow_fast<false, true>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_trace_shadow_fast<true, false>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_trace_shadow_fast<true, true>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_trace_shadow_fast_inst<false, false>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_trace_shadow_fast_inst<false, true>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_trace_shadow_fast_inst<true, false>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_trace_shadow_fast_inst<true, true>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_tail<MASK_TAIL, false>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_tail<MASK_TAIL, true>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_tail<(MASK_TAIL | FEAT_QMC), false>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_tail<(MASK_TAIL | FEAT_QMC), true>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_tail<MASK_COAT, false>(DeviceScene, PathState, PassParams, uint32_t);
extern template __global__ void k_tail<(MASK_COAT | FEAT_QMC), false>(DeviceScene, PathState, PassParams, uint32_t);

// CubemapCamera's layout tables (cameras/CubemapCamera.cpp:10-46), indexed by projection mode and face
__device__ const int g_cubeResU[4] = {4, 3, 6, 1}, g_cubeResV[4] = {3, 4, 1, 6};
__device__ const int g_cubeOffsetU[4][6] = {{2, 0, 1, 1, 1, 3}, {1, 1, 1, 1, 0, 2}, {0, 1, 2, 3, 4, 5}, {0, 0, 0, 0, 0, 0}};
__device__ const int g_cubeOffsetV[4][6] = {{1, 1, 0, 2, 1, 1}, {1, 3, 0, 2, 1, 1}, {0, 0, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5}};
__device__ const int g_cubeBasisU[4][6] = {{5, 4, 0, 0, 0, 1}, {5, 5, 5, 5, 0, 1}, {5, 4, 0, 0, 0, 1}, {5, 4, 0, 0, 0, 1}};     // indices of +x -x +y -y +z -z
__device__ const int g_cubeBasisV[4][6] = {{3, 3, 4, 5, 3, 3}, {3, 2, 0, 1, 3, 3}, {3, 3, 4, 5, 3, 3}, {3, 3, 4, 5, 3, 3}};
PT_DEV f3 cubeBasis(int i) { const float sgn = (i & 1) ? -1.0f : 1.0f; return mk3((i >> 1) == 0 ? sgn : 0.0f, (i >> 1) == 1 ? sgn : 0.0f, (i >> 1) == 2 ? sgn : 0.0f); }

// cameras/EquirectangularCamera.cpp (TGHIP_CAMERA_EQUIRECTANGULAR): the full sphere around the camera's position.  nextPath writes every fresh camera path
// with the pinhole's ray; for this camera the direction is replaced here, in a launch in front of every closest-hit launch, for the slots of the
// workgroup's Q_EXTP (the freshly generated camera rays) -- from the same two random numbers: the path's stream restarted at (seed, pixel, sample) or
// at Sobol' dimension 0, as nextPath drew them (EquirectangularCamera::sampleDirection draws its filter offset exactly where the pinhole's does,
// :70-83).  Keeps the camera's sine and cosine out of every kernel that holds nextPath -- the metric's closest-hit kernel among them.  Idempotent.
__global__ __launch_bounds__(256) void k_camera_rays(DeviceScene s, PathState st, PassParams pp)
{
    const uint32_t W = st.slots_per_block >> 5;
    const uint32_t first = blockIdx.x*st.slots_per_block;
    CameraRef cam = *asConst(s.camera);
    for (uint32_t local = threadIdx.x; local < st.slots_per_block; local += blockDim.x) {
        const uint32_t word = st.bm[(uint32_t)Q_EXTP*st.bmStride + blockIdx.x*W + (local >> 5)];
        if (!((word >> (local & 31u)) & 1u))
            continue;
        const uint32_t slot = first + local;
        const uint4 misc = slotU4(st, A_MISC, slot), sm = slotU4(st, A_SAMP, slot);
        const uint32_t pixel = misc.z, px = pixel % pp.width, py = pixel/pp.width;
        Rng rng = rngStart(pp.seed, pixel, sm.x);
        if (pp.flags & TGHIP_PASS_SOBOL)
            rngStartSobol(rng, s, pp, px, py, pixel, sm.x, 0u);
        const float xi0 = rngNext1DT<true>(rng), xi1 = rngNext1DT<true>(rng);
        float fu = 0.0f, fv = 0.0f;
        if (cam.filter_type == TGHIP_FILTER_BOX) { fu = xi0 - 0.5f; fv = xi1 - 0.5f; }
        else if (cam.filter_type == TGHIP_FILTER_TABULATED) { fu = filterSample1D(cam, xi0); fv = filterSample1D(cam, xi1); }
        f3 l;
        bool ok = true;
        if (cam.type == TGHIP_CAMERA_CUBEMAP) {
            // CubemapCamera::sampleDirection / uvToFace / uvToDirection / faceToDirection (cameras/CubemapCamera.cpp:153-166, 95-112, 74-80) with the tables of
            // :10-46 and prepareForRender (:217-232); blade_count = the projection mode.  A pixel outside the six faces: the sample fails (black, PathTracer.cpp:27-28)
            const int mode = cam.blade_count;
            const float faceW = 1.0f/(float)g_cubeResU[mode], faceH = 1.0f/(float)g_cubeResV[mode];
            float u = ((float)px + 0.5f)*cam.pixel_size_x, v = ((float)py + 0.5f)*cam.inv_xf[9];
            int face = -1;
            for (int i = 0; i < 6 && face < 0; ++i) {
                const float dx = u - (float)g_cubeOffsetU[mode][i]*faceW, dy = v - (float)g_cubeOffsetV[mode][i]*faceH;
                if (dx >= 0.0f && dy >= 0.0f && dx <= faceW && dy <= faceH) face = i;
            }
            ok = face >= 0;
            if (ok) {
                u += fu*cam.pixel_size_x; v += fv*cam.inv_xf[9];
                const float dx = u - (float)g_cubeOffsetU[mode][face]*faceW, dy = v - (float)g_cubeOffsetV[mode][face]*faceH;
                const float ox = dx/faceW, oy = dy/faceH;
                const f3 b = cubeBasis(face), bu = cubeBasis(g_cubeBasisU[mode][face]), bv = cubeBasis(g_cubeBasisV[mode][face]);
                l = normalized(b + bu*(ox*2.0f - 1.0f) + bv*(oy*2.0f - 1.0f));
            } else {
                l = mk3(0.0f, 0.0f, 1.0f);
            }
        } else {
        // uvToDirection (:26-36); inv_xf holds _rot and 1 / res_y (include/tungsten_hip.h)
        const float u = ((float)px + 0.5f + fu)*cam.pixel_size_x, v = ((float)py + 0.5f + fv)*cam.inv_xf[9];
        const float phi = (u - 0.5f)*PT_TWO_PI, theta = (1.0f - v)*PT_PI;
        const float sinTheta = sinfH(theta);
        l = mk3(cosfH(phi)*sinTheta, -cosfH(theta), sinfH(phi)*sinTheta);
        }
        const f3 d = mk3(cam.inv_xf[0]*l.x + cam.inv_xf[1]*l.y + cam.inv_xf[2]*l.z + 0.0f,      // Mat4f*Vec3f: the (zero) translation column is added
                         cam.inv_xf[3]*l.x + cam.inv_xf[4]*l.y + cam.inv_xf[5]*l.z + 0.0f,
                         cam.inv_xf[6]*l.x + cam.inv_xf[7]*l.y + cam.inv_xf[8]*l.z + 0.0f);
        if (ok) {
            const float4 rd = slotF4(st, A_RAY_D, slot);
            slotF4(st, A_RAY_D, slot) = mk4(d, rd.w);
        } else {                                 // as nextPath leaves a failed camera sample: a ray that can hit nothing, no throughput
            slotF4(st, A_RAY_D, slot) = mk4(d, -1.0f);
            const float4 thr = slotF4(st, A_THR, slot);
            slotF4(st, A_THR, slot) = make_float4(0.0f, 0.0f, 0.0f, thr.w);
        }
    }
}


// =============================================================================================
// Host-side shim
// =============================================================================================

namespace {

std::mutex g_errMutex;
std::string g_createError = "no error";

// The streams of a context, kept per device for the life of the process and handed from a destroyed context to the next one created:
// HIP binds a stream to one of a few hardware queues when it is created (GPU_MAX_HW_QUEUES, 4 by default), and the streams of a context
// created after another one was destroyed came out with a binding on which the four parts of the wavefront loop no longer ran side by
// side (materialtest 1280x720x256: 840 Msamples/s in a process's first context, 675 in every later one; the 16-spp passes of the
// as-shipped scene 24 ms against 35 ms).  A renderer that is opened once per frame or per scene keeps its first context's streams this way.
struct StreamSet {
    hipStream_t main = nullptr, part[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, abort = nullptr;
};
std::mutex g_streamMutex;
std::map<int, std::vector<StreamSet>> g_freeStreams;   // device ordinal -> sets no context holds (never destroyed: they live as long as the process)

struct DeviceBuffers {
    std::vector<void *> allocs;
    ~DeviceBuffers() { release(); }
    void release() { for (void *p : allocs) (void)hipFree(p); allocs.clear(); }
};

} // namespace

struct tghip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t partStream[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // streams of parts 1..3 of the split wavefront loop ("streams" option)
    hipStream_t classStream[8][2] = {};   // per part: the streams of the shading classes that run beside the part's own ("class_streams" option)
    hipEvent_t evFork[8] = {}, evJoin[8][2] = {};
    bool shortBatch = false;              // the pass being rendered does not fill the pool once: one stream, 4 workgroups per CU (tghip_render_pass)
    int instSimpleOpt = 1;                // "inst_simple": classes 0 / 2 of instanced scenes on the MASK_SIMPLE_INST variant instead of MASK_FULL
    int classStreamsOpt = 0;              // measured: 735-800 Msamples/s against 825-830 with the classes one after the other on the part's stream
    hipStream_t launchStream = nullptr;   // where the launch helpers put their kernels (stream, or the stream of the part being launched)
    hipEvent_t evPart[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, evMain = nullptr;
    hipEvent_t evRot[8] = {};             // "rotate_streams": the end of a part's last iteration (the part's next one runs on another stream)
    bool rotateStreamsOpt = false;        // "rotate_streams" option (runBatch)
    int streamsOpt = 0;                   // "streams": 1 .. 4 parts of the pool on as many streams, 0 = the measured default (four for single-level
                                          // BVH scenes; instanced scenes lose 2.5 % with two)
    hipDeviceProp_t prop;
    std::string error = "no error";

    // scene
    bool haveScene = false;
    DeviceBuffers sceneMem;
    DeviceScene scene;
    int bvhDepth = 0;
    int wideDepth = 0;                    // levels of the 8-wide BVH (0: the scene has none, the kernels walk the BVH2)
    bool wideOpt = true;                  // "wide_bvh" option: use it when the scene carries one
    // "wide_closest" / "wide_shadow": which kernels walk the wide BVH when the scene has one.  -1 = the measured default:
    // both, except that closest-hit rays of instanced scenes stay on the two-level BVH2 kernel (instances10k 1080p, one MI355X:
    // closest-hit 849 us per launch on the BVH2 against 1061 us on the wide tree -- every instance entered costs the wide
    // walk extra turns --, shadow rays 905 against 517 us)
    int instDynOpt = 1;                   // "inst_dyn": closest-hit rays of instanced scenes on the dynamic-fetch kernel (k_trace_closest_inst) instead of the static one
    int instWideOpt = 1;                  // "inst_wide": ... with the masters' subtrees walked through the wide BVH (k_trace_closest_instw); 0 = the BVH2 walk of round 5
    int instPhaseMin = 24;                // "inst_phase_min" / "inst_refill_at" (PathState::inst_*)
    int instRefillAt = 48;
    int bvhMasterDepth = 0;               // instanced scenes: the part of bvhDepth that is the deepest master's BVH2 subtree, and the wide walk's levels inside a master
    int wideMasterDepth = 0;
    int wideClosestOpt = -1, wideShadowOpt = -1;
    // "tail_kernel" / "tail_threshold": once a host check finds at most tail_threshold paths alive in the pool, the rest of the batch runs in
    // k_tail (one launch per part: every workgroup iterates over its own slots until they are done).  The kernel is built for latency, not
    // throughput (one shading variant for every class, one wave per SIMD): measured, Msamples/s for thresholds off / 2 Ki / 8 Ki / 32 Ki / 128 Ki:
    // mesh1m 605 / 625 / 611 / 575 / 514, materialtest 963 / 966 / 964 / 968 / 966, materialtest as shipped 573 / 611 / 630 / 623 / 625
    bool mediaSimple = false;             // a media scene whose surface BSDFs MASK_MEDIA covers (no instances, no mesh emitters)
    bool mediaLeanOpt = true;             // "media_lean": shade such scenes with k_shade<MASK_MEDIA> instead of <BSDF_MASK_ALL>
    bool foldFinishOpt = true;            // "fold_finish"
    bool finishLeanOpt = false;           // "finish_lean" = 1: the folded finish of flag-less passes through nextPath's lean variant -- measured SLOWER (profiles/r6_ab_finish_lean.txt), off
    bool topTreeOpt = true;               // "top_tree": 0 = ignore TgHipSceneDesc::top_nodes at the next upload (flat lists walked in record order: faster, not the reference's ties)
    bool mergeMissOpt = true;             // "merge_miss"
    bool tailOpt = true;
    long long tailThreshold = 8192;
    int shadeLdsPad = 0;                  // "shade_lds_pad": bytes of unused dynamic LDS per k_shade workgroup (an occupancy throttle for experiments: profiles/r6_ab_shade_occupancy.txt)
    bool instShadowFast = true;           // "inst_shadow_fast": instanced scenes' shadow rays on k_trace_shadow_fast_inst (0: k_trace_shadow_wide<., ., INST>)
    bool instShadowJoin = true;           // "inst_shadow_join": 0 = the instanced wide shadow kernel without PT_TURN_JOIN (the miscompiled variant; repro tool only)
    bool failReduce = false;              // "fail_reduce" option (fault injection for the reduce's callers)
    int wideStride = int(PT_WIDE_NODE_BYTES);   // bytes per device node: 128 (pt_kernels.h: PT_WIDE_HALF); the byte layout also takes 80 ("wide_node_stride" option, at the next upload)
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
    uint32_t poolWalkArrays = 0;          // walk arrays the pool carries behind the A_* ones (0: walks are never suspended)
    uint32_t poolWalkWanted = 0;          // ... and how many the scene it was laid out for asked for (more than fit the 32-bit offsets: none)
    uint32_t *hostLive = nullptr;         // pinned mirror of PathState::live for the loop condition
    unsigned long long walkStats[2][PT_WALK_STATS] = {{0}, {0}};   // BlockStats::walk summed over workgroups and folds (tghip_get_walk_stats)
    std::vector<BlockCtl> hostCtl;        // scratch for tghip_get_counters
    std::vector<BlockStats> hostStats;

    // options
    // path pool size.  Every kernel of an iteration costs a fixed 140-180 us on top of the time that grows with its rays -- the
    // longest walks / the latency of one shading turn, measured by doubling the pool: materialtest closest-hit 246 -> 348 us, shadow
    // 347 -> 518 us, k_shade 421 -> 665 us per launch for twice the rays -- so the pool is as large as the 32-bit slot offsets allow
    // (8 M slots x 272 B = 2.2 GB of the 288 GB): 1280x720x256 669 -> 745 (4 M) -> 818 (8 M) Msamples/s, mesh1m 404 -> 472 -> 507
    long long maxSlots = 1ll << 23;
    int slotsPerBlockOpt = 0;             // "slots_per_block": upper bound on the slots of one workgroup (0 = PT_MAX_SLOTS_PER_BLOCK)
    bool maxSlotsSet = false;             // "max_slots" was given explicitly
    long long maxItems = 1ll << 26;       // work items per batch (partial-sum buffer = 16 B each)
    int chunkSamples = 0;                 // "chunk_samples": samples per work item; 0 = 4, or fewer when the pass is short (tghip_wait)
    size_t partialCap = 0;
    float4 *partial = nullptr;

    // shading classes of the uploaded scene (rec_class) and the kernel variants chosen for them
    bool haveComplex = false;             // some primitive record uses a BSDF of class 1, 2 or 3
    uint32_t complexMask = 0;             // union of the BSDF types inside those materials
    bool classPresent[PT_NUM_CLASSES] = {false, false, false, false};   // shading classes (pt_kernels.h: PT_NUM_CLASSES) that occur among the records
    uint32_t classMask[PT_NUM_CLASSES] = {0, 0, 0, 0};                  // ... and the BSDF types inside each
    bool haveForward = false;             // some BSDF has a forward lobe (shadow rays attenuate instead of stop)
    bool haveMeshLight = false;           // a triangle mesh is a sampled light: closest-hit shadow walk, MASK_FULL shading
    bool thinlens = false;                // thin-lens camera: passes run the EXT kernel variants (PT_PASS_THINLENS)
    bool cameraFix = false;               // equirectangular camera: k_camera_rays rewrites the fresh camera rays before they are traced
    bool haveSolids = false;              // cube / sphere / disk records: the dynamic-fetch kernels' SOLIDS variants
    TgHipAuxPixel *dAux = nullptr;        // auxiliary output buffers (allocated by the first TGHIP_PASS_AUX pass)
    float *dSamples = nullptr;            // TGHIP_PASS_SAMPLES: per-sample radiance of the last such pass
    void *rankComm = nullptr;             // tghip_comm_init_rank: this process's ncclComm_t (one process per GPU); destroyed with the context
    int rankCount = 0, rankIndex = 0;
    float *redSum = nullptr;              // tghip_reduce_framebuffers: where the reduced image lands when this context is the root
    uint32_t *redCount = nullptr;
    size_t redCap = 0;                    // pixels redSum / redCount were allocated for (a re-upload may change the resolution)
    // the tiles of the shard the last pass rendered (tghip_tile_owner), on the host and on the device; key = {W, H, shard index, shard count}
    std::vector<uint32_t> hostOwnedTiles;
    uint32_t *dOwnedTiles = nullptr;
    size_t ownedCap = 0;
    uint32_t ownedKey[4] = {0, 0, 0, 0};
    size_t samplesCap = 0, samplesFloats = 0;
    bool auxPass = false;                 // the pass being rendered keeps them: BSDF_MASK_ALL shading, no fused / dynamic-fetch shadow kernels
    int thrShadeAll = 256;                // workgroup size of k_shade<BSDF_MASK_ALL> (media scenes, TGHIP_PASS_AUX passes)
    bool haveCylinder = false;            // cylinder primitives -- or a bump-mapped bsdf (TgHipBsdf::bump1): BSDF_MASK_ALL shading (the only FEAT_CYLINDER /
                                          // FEAT_BUMP variant), never fused
    bool haveMedia = false;               // participating media: BSDF_MASK_ALL shading (the only FEAT_MEDIA variant), closest-hit shadow walk, never fused
    bool haveInstances = false;           // instance records: two-level traversal kernels (INST), MASK_FULL shading, never the flat list
    bool hoistOpt = true;                 // "hoist_quad": the scene's one quad tested before the decoupled walks instead of inside them (at the next upload)
    bool tailFamilyOpt = true;            // "tail_family": k_tail<MASK_COAT> for scenes without class-2 / class-3 materials and without solids
    bool tablesFit = true;                // objects + bsdfs + textures + light lists fit the shading workgroups' LDS copy (pt_kernels.h: stageSceneTables)
    bool tablesFitScene = true;           // ... as decided at upload; "lds_tables" = 0 shades as if they did not (the GLOBAL_TABLES variant: tests)
    int hoistedRecScene = -1;             // DeviceScene::hoisted_rec as decided at upload ("hoist_quad" switches it at run time: the skip word stays in the node)
    int envTexScene = -1;                 // the sampled environment map whose marginal tables ride in LDS; "env_lds" = 0 samples it through its global tables
    bool leanScene = false;               // no bitmap texture, no infinite light, <= 1 sampled light, no triangles: k_shade<MASK_LEAN>
    bool countTraversal = false;
    int checkInterval = 0;                // "check_interval": wavefront iterations between host-side liveness checks; 0 = 16 for batches that refill
                                          // the pool several times (every check drains all streams: materialtest 825 / 835 / 845 Msamples/s for
                                          // 4 / 8 / 16), 4 for short ones (the iterations after the last path ended are wasted)
    int blocksPerCuOpt = 0;               // "blocks_per_cu" option; 0 = auto (see chooseThreads)
    int blocksPerCu = 4;                  // persistent workgroups per CU (the same grid for every kernel of a pass)
    int gridRounds = 1;                   // "grid_rounds": launch this many times the resident workgroups (each owns 1/rounds of the slots);
                                          // the dispatcher starts the later ones as the first finish, filling the drain tail of a launch
    // threads per workgroup, per kernel: chosen at upload so that `blocksPerCu` workgroups of EVERY kernel are
    // resident at once (no second scheduling round), i.e. each kernel runs at its own best occupancy on one grid
    int thrClosest = 256, thrShadow = 256, thrShadeSimple = 192, thrShadeComplex = 128;
    int thrOverride[4] = {0, 0, 0, 0};
    bool loopOpt = true;                  // "run_to_completion": fused flat-list scenes whose materials are all of class 0 render in ONE launch
    bool fuseFlatOpt = true;              // "fuse_flat": flat-list scenes without forward lobes trace + shadow-test inside k_shade
    // "suspend_lanes" / "suspend_turns" / "suspend_min_queue" (PathState::suspend_*): walk time-slicing of the wide traversal kernels
    int suspendLanes = 12, suspendTurns = 16, suspendMinQueue = 1024;   // (measured, profiles/README.md: materialtest +0.5 %, mesh1m +4 % over none; round 5 on the final kernels, r5_sweep_final_kernels.txt: 12 lanes +0.5 % / +1 % over 16, 8 lanes +0.8 % / -3 %)
    int ldsNodesOpt = 0;                  // "lds_nodes": nodes of the top of the wide tree kept in LDS by those kernels (9 / 73 / 585 = two / three / four levels; measured: no gain)
    uint32_t numWideNodes = 0;
    int decoupleOpt = 1;                  // "decouple": the wide kernels of single-level scenes test a record AND visit a node per turn (k_trace_closest_wide<.., DECOUPLED>)
    int leafBatch = 1;                    // "leaf_batch" (PathState::leaf_batch)
    int leafBatchBvh2 = 0;                // "leaf_batch_bvh2" (PathState::leaf_batch_bvh2); 0 = leaf_batch, or the measured value for two-level scenes
    bool poolRecords = false;             // "pool_layout" option: 1 = slot records (PathState::records)
    long long poolPad = 9472;             // bytes between the per-slot arrays of the pool (multiple of 16)
    bool dynamicFetch = true;             // BVH scenes: closest-hit kernel with dynamic ray fetch (k_trace_closest_dyn)
    bool timeKernels = false;             // HIP events around every launch of the wavefront loop (bench.py roofline)
    std::vector<hipEvent_t> evPool;

    // abort (PathTraceIntegrator::abortRender): a host-side request flag, mirrored into one device word the kernels poll.
    // The word is allocated once per context (never with the reallocatable pool), so tghip_abort may write it from any
    // thread at any time; the request is cleared by tghip_render_pass only, so one that lands before the pass's kernels
    // have started is not lost.
    std::atomic<bool> abortRequested{false};
    uint32_t *abortFlagDev = nullptr;
    hipStream_t abortStream = nullptr;
    std::mutex abortMutex;                // serialises tghip_abort callers (they share abortStream)

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
static int uploadArray(tghip_ctx *ctx, DeviceBuffers &mem, const T *src, size_t count, P *dst, size_t padBytes = 0)   // P = (restrict-qualified) const T *
{
    size_t bytes = std::max<size_t>(count, 1)*sizeof(T) + padBytes;   // (padBytes: readable, uninitialised room behind the last element)
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
    if (b.type == TGHIP_BSDF_SMOOTH_COAT || b.type == TGHIP_BSDF_ROUGH_COAT || b.type == TGHIP_BSDF_TRANSPARENCY)
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
    if ((b.type == TGHIP_BSDF_ROUGH_CONDUCTOR || b.type == TGHIP_BSDF_ROUGH_DIELECTRIC || b.type == TGHIP_BSDF_ROUGH_PLASTIC || b.type == TGHIP_BSDF_ROUGH_COAT) &&
        b.distribution == TGHIP_DIST_PHONG)
        m |= FEAT_PHONG;                     // outside every family mask: such a material is shaded by the full variant (pt_scene.h: mfDist)
    if (b.type == TGHIP_BSDF_SMOOTH_COAT || b.type == TGHIP_BSDF_ROUGH_COAT || b.type == TGHIP_BSDF_TRANSPARENCY)
        m |= bsdfTypeMask(s, b.sub0, depth + 1);
    if (b.type == TGHIP_BSDF_MIXED)
        m |= bsdfTypeMask(s, b.sub0, depth + 1) | bsdfTypeMask(s, b.sub1, depth + 1);
    return m;
}

// Depth of the subtree under `root` (also validates child references).  `level`: 0 = the scene's tree (leaves: non-instance records and
// instance-set records, which are collected in `found`), 1 = the reference's tree behind a set record (leaves: one or two slots of
// inst_prims; the instance records behind them are collected in `found`), 2 = a master's subtree (triangles and the like only).
static int subtreeDepth(const TgHipSceneDesc *s, int32_t root, size_t &visited, int level, std::vector<uint32_t> *found)
{
    std::vector<std::pair<int32_t, int>> stack;
    stack.emplace_back(root, 1);
    int depth = 0;
    while (!stack.empty()) {
        auto cur = stack.back();
        stack.pop_back();
        if (cur.first < 0) {
            uint32_t first = TGHIP_LEAF_FIRST(cur.first), count = TGHIP_LEAF_COUNT(cur.first);
            if (level == 1) {
                if (count < 1 || count > 2 || first + count > s->num_inst_prims) return -1;
                for (uint32_t k = first; k < first + count; ++k) {
                    const uint32_t ri = s->inst_prims[k];
                    if (ri >= s->num_top_recs || TGHIP_REC_KIND(s->recs[ri].meta) != TGHIP_REC_INSTANCE) return -1;
                    found->push_back(ri);
                }
                continue;
            }
            if (first + count > s->num_recs) return -1;
            for (uint32_t i = first; i < first + count; ++i) {
                const uint32_t kind = TGHIP_REC_KIND(s->recs[i].meta);
                if (kind == TGHIP_REC_INSTANCE) return -1;           // instance records are reached through their set's tree only
                if (kind == TGHIP_REC_INSTANCE_SET) {
                    if (level != 0 || count != 1) return -1;
                    found->push_back(i);
                }
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

// Stack words the BVH2 traversal needs: the scene's tree; with `instances` primitives, above it the reference's tree over the instances
// and the deepest master subtree (pt_kernels.h: instanceSetIntersect).
static int bvhDepthOf(const TgHipSceneDesc *s, int *masterDepthOut = nullptr)
{
    if (masterDepthOut) *masterDepthOut = 0;
    size_t visited = 0;
    std::vector<uint32_t> sets;
    int depth = subtreeDepth(s, 0, visited, 0, &sets);
    if (depth < 0 || (sets.empty() != (s->num_instances == 0))) return -1;
    if (sets.empty()) return depth;
    if (!s->inst_prims || !s->inst_leaf_boxes) return -1;
    std::vector<uint32_t> inst;
    int ref = 0;
    for (uint32_t set : sets) {
        int32_t root;
        std::memcpy(&root, &s->recs[set].c[0], 4);
        if (root == 0) return -1;
        int d = subtreeDepth(s, root, visited, 1, &inst);
        if (d < 0) return -1;
        ref = std::max(ref, d);
    }
    if (inst.size() != s->num_instances) return -1;
    std::vector<uint32_t> roots;
    for (uint32_t i : inst) {
        uint32_t root, leaf;
        std::memcpy(&root, &s->recs[i].c[0], 4);
        std::memcpy(&leaf, &s->recs[i].c[1], 4);
        if (root == 0 || root >= s->num_nodes || leaf >= s->num_inst_prims) return -1;
        roots.push_back(root);
    }
    std::sort(roots.begin(), roots.end());
    roots.erase(std::unique(roots.begin(), roots.end()), roots.end());
    int master = 0;
    for (uint32_t root : roots) {
        int d = subtreeDepth(s, int32_t(root), visited, 2, nullptr);
        if (d < 0) return -1;
        master = std::max(master, d);
    }
    if (masterDepthOut) *masterDepthOut = master;
    return depth + ref + master + 3;
}

// Validates the wide BVH -- the top-level tree from node 0 and, with instances, the masters' subtrees behind it (roots in the
// instance records): children behind their parent, every node in one tree, record runs inside the record array -- and returns
// the stack depth the walk needs (-1 when malformed).
static int wideDepthOf(const TgHipSceneDesc *s, int *masterDepthOut = nullptr)
{
    if (masterDepthOut) *masterDepthOut = 0;
    const uint32_t n = s->num_wide_nodes;
    std::vector<uint8_t> depth(n, 0);
    depth[0] = 1;
    uint32_t firstMaster = n;
    for (uint32_t i = 0; i < s->num_recs && s->num_instances; ++i) {
        if (TGHIP_REC_KIND(s->recs[i].meta) != TGHIP_REC_INSTANCE) continue;
        uint32_t root;
        std::memcpy(&root, &s->recs[i].c[2], 4);
        if (root == 0 || root >= n) return -1;
        depth[root] = 1;
        firstMaster = std::min(firstMaster, root);
    }
    int topDepth = 1, masterDepth = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const TgHipWideNode &w = s->wide_nodes[i];
        if (depth[i] == 0) return -1;                        // unreachable node: not a forest in breadth-first order
        const uint32_t kids = uint32_t(__builtin_popcount(w.imask));
        if (kids && (w.child_base <= i || uint64_t(w.child_base) + kids > n)) return -1;
        if (kids && i < firstMaster && w.child_base + kids > firstMaster) return -1;   // the top level does not reach into a master
        for (uint32_t k = 0; k < kids; ++k) {
            if (depth[w.child_base + k] != 0) return -1;     // two parents
            depth[w.child_base + k] = uint8_t(depth[i] + 1);
        }
        if (i < firstMaster) topDepth = std::max(topDepth, int(depth[i]) + (kids ? 1 : 0));
        else masterDepth = std::max(masterDepth, int(depth[i]) + (kids ? 1 : 0));
        if (topDepth > TGHIP_MAX_WIDE_DEPTH || masterDepth > TGHIP_MAX_WIDE_DEPTH) return -1;
        for (int sl = 0; sl < 8; ++sl) {
            const uint32_t bits = (w.leaf_valid >> (4*sl)) & 15u;
            if ((bits & (bits + 1u)) != 0u || (bits && (w.imask & (1u << sl)))) return -1;   // records 0 .. count-1 of a leaf slot
        }
        const uint32_t recLimit = i < firstMaster ? (s->num_top_recs ? s->num_top_recs : s->num_recs) : s->num_recs;
        if (w.leaf_valid && uint64_t(w.rec_base) + uint32_t(__builtin_popcount(w.leaf_valid)) > recLimit) return -1;
        for (int a = 0; a < 3; ++a)
            if (w.exp[a] == 0 || w.exp[a] == 255) return -1;
    }
    const int total = s->num_instances ? topDepth + masterDepth + 3 : topDepth;   // + what entering an instance parks on the stack
    if (masterDepthOut) *masterDepthOut = masterDepth;
    return total > TGHIP_MAX_WIDE_DEPTH ? -1 : total;
}

static bool isFlat(const tghip_ctx *ctx) { return ctx->scene.num_recs <= TGHIP_FLAT_MAX_RECS && !ctx->haveInstances; }
// the single-level traversal kernels walk the 8-wide BVH when the scene carries one
static bool useWide(const tghip_ctx *ctx) { return ctx->wideDepth > 0 && ctx->wideOpt && ctx->dynamicFetch && !isFlat(ctx); }


static bool wideClosest(const tghip_ctx *ctx);
static bool wideShadowRays(const tghip_ctx *ctx);
// Most slots a workgroup of the uploaded scene owns (the pool is sized by it, and the traversal kernels' LDS queue area): 4096 where both
// traversal kernels are the wide ones.  The static-fetch BVH2 kernels of flat-list and instanced scenes keep a thread's queue entries in
// 16 half registers (OrderRegs: <= 16 x threads), and their deep stacks leave less LDS: 2048, as measured (instances10k: 4096 slots with
// the smaller workgroups that then fit are not faster).
static uint32_t slotCap(const tghip_ctx *ctx)
{
    const bool allWide = !isFlat(ctx) && !ctx->haveInstances && wideClosest(ctx) && wideShadowRays(ctx) && !ctx->haveForward && !ctx->haveMeshLight;
    uint32_t cap = allWide ? PT_MAX_SLOTS_PER_BLOCK : 2048u;      // (BlockLdsSmall in the other traversal kernels)
    if (ctx->slotsPerBlockOpt > 0)
        cap = std::min<uint32_t>(cap, uint32_t(ctx->slotsPerBlockOpt));
    return std::max(cap, 64u);
}

// Dynamic LDS of the traversal kernels: one node stack of bvhDepth ints per thread (a root-to-leaf walk pushes at
// most one far child per internal level), aliased with the expanded queue (2 B per slot) that is consumed before
// traversal starts.  Flat-list scenes need no stack.
static size_t traceLdsBytes(const tghip_ctx *ctx, int threads)
{
    const bool flat = isFlat(ctx);
    size_t stack = flat ? 0 : size_t(std::max(ctx->bvhDepth, 1))*size_t(threads)*sizeof(int);
    return std::max<size_t>(stack, size_t(slotCap(ctx))*sizeof(unsigned short));
}

// dynamic-fetch traversal kernels keep the expanded queue next to the stacks
static size_t dynLdsBytes(const tghip_ctx *ctx, int threads)
{
    return size_t(slotCap(ctx))*sizeof(unsigned short) + size_t(std::max(ctx->bvhDepth, 1))*size_t(threads)*sizeof(int);
}

// the wide kernels: expanded queue + one 8-byte group entry per tree level and thread
// nodes of the top of the wide tree the DECOUPLED kernels keep in LDS (PathState::lds_nodes): whole levels of the breadth-first array
static uint32_t ldsNodeCount(const tghip_ctx *ctx)
{
    if (!PT_LDS_TOP || !ctx->decoupleOpt || ctx->haveInstances || ctx->ldsNodesOpt == 0) return 0u;   // (PT_LDS_TOP = 0, the product: the option is accepted and has no effect)
    const uint32_t n = ctx->scene.wide ? ctx->numWideNodes : 0u;
    return std::min<uint32_t>(n, uint32_t(ctx->ldsNodesOpt));
}
static size_t wideLdsBytes(const tghip_ctx *ctx, int threads)
{
    return size_t(slotCap(ctx))*sizeof(unsigned short) + size_t(std::max(ctx->wideDepth, 1))*size_t(threads)*sizeof(uint2)
         + size_t(ldsNodeCount(ctx))*size_t(ctx->wideStride);
}

// k_trace_closest_instw: expanded queue + the BVH2 stack of the scene's tree and the reference's tree over the instances (no master on it) + the
// masters' group stack (8-byte entries, 8-byte aligned)
static int instTreeDepth(const tghip_ctx *ctx) { return std::max(ctx->bvhDepth - ctx->bvhMasterDepth, 1); }
static size_t instWideLdsBytes(const tghip_ctx *ctx, int threads)
{
    const size_t ints = ((size_t(slotCap(ctx)) >> 1) + size_t(instTreeDepth(ctx))*size_t(threads) + 1u) & ~size_t(1);
    return ints*sizeof(int) + size_t(std::max(ctx->wideMasterDepth, 1))*size_t(threads)*sizeof(uint2);
}
static bool instWide(const tghip_ctx *ctx) { return ctx->haveInstances && ctx->instWideOpt && ctx->wideDepth > 0 && ctx->wideMasterDepth > 0 && ctx->wideOpt; }
static bool wideClosest(const tghip_ctx *ctx) { return useWide(ctx) && (ctx->wideClosestOpt < 0 ? !ctx->haveInstances : ctx->wideClosestOpt != 0); }
static bool wideShadowRays(const tghip_ctx *ctx) { return useWide(ctx) && ctx->wideShadowOpt != 0; }

static int launchGrid(const tghip_ctx *ctx) { return ctx->prop.multiProcessorCount*std::max(ctx->blocksPerCu, 1)*std::max(ctx->gridRounds, 1); }

// Largest workgroup size (multiple of 64, <= maxThreads) at which `blocksPerCu` workgroups of `kernel` fit on a CU.
template<typename K>
static int pickThreads(const tghip_ctx *ctx, K kernel, int maxThreads, int ldsMode)   // 0: no dynamic LDS, 1: traceLdsBytes, 2: dynLdsBytes, 3: wideLdsBytes, 4: instWideLdsBytes
{
    for (int t = maxThreads; t >= 128; t -= 64) {
        int nb = 0;
        size_t lds = ldsMode == 1 ? traceLdsBytes(ctx, t) : ldsMode == 2 ? dynLdsBytes(ctx, t) : ldsMode == 3 ? wideLdsBytes(ctx, t) : ldsMode == 4 ? instWideLdsBytes(ctx, t) : 0;
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
        if (ctx->countTraversal) {
            ctx->counters.nodes_visited += t.nodes_visited; ctx->counters.prims_tested += t.prims_tested;
            ctx->counters.nodes_visited_shadow += t.nodes_visited_shadow; ctx->counters.prims_tested_shadow += t.prims_tested_shadow;
        }
    }
    if (ctx->countTraversal)
        for (int k = 0; k < 2; ++k)
            for (size_t b = 0; b < g; ++b)
                for (int i = 0; i < PT_WALK_STATS; ++i) {
                    if (i == 11) ctx->walkStats[k][i] = std::max(ctx->walkStats[k][i], ctx->hostStats[b].walk[k][i]);
                    else ctx->walkStats[k][i] += ctx->hostStats[b].walk[k][i];
                }
    if (std::getenv("TGHIP_VERBOSE") && ctx->countTraversal) {
        for (int k = 0; k < 2; ++k) {
            unsigned long long t[12] = {0};
            for (size_t b = 0; b < g; ++b) for (int i = 0; i < 12; ++i) { if (i == 11) t[i] = std::max(t[i], ctx->hostStats[b].walk[k][i]); else t[i] += ctx->hostStats[b].walk[k][i]; }
            if (!t[4]) continue;
            const double waves = double(t[4]), us = 0.01;
            std::fprintf(stderr, "[tghip] %s walk, per wave launch: expand %.1f us | loop with queue %.1f us (%.1f turns, %.1f busy lanes) | dry %.1f us (%.1f turns, %.1f busy lanes) | wait+write-back %.1f us | "
                                 "longest loop %.1f us | suspended %llu resumed %llu walks in %llu wave launches\n", k == 0 ? "closest-hit" : "shadow",
                         double(t[0])*us/waves, double(t[1])*us/waves, double(t[5])/waves, t[5] ? double(t[7])/double(t[5]) : 0.0,
                         double(t[2])*us/waves, double(t[6])/waves, t[6] ? double(t[8])/double(t[6]) : 0.0, double(t[3])*us/waves, double(t[11])*us, t[9], t[10], t[4]);
        }
    }
    if (std::getenv("TGHIP_VERBOSE")) {
        unsigned long long turns = 0, dry = 0;
        for (size_t b = 0; b < g; ++b) { turns += ctx->hostStats[b].prof[10]; dry += ctx->hostStats[b].prof[11]; }
        if (turns) std::fprintf(stderr, "[tghip] closest-hit walk: %llu wave turns (%llu after the queue ran dry)\n", turns, dry);
    }
#ifdef PT_PROFILE
    {
        unsigned long long tot[16] = {0};
        for (size_t b = 0; b < g; ++b) for (int k = 0; k < 16; ++k) if (k != 10 && k != 11) tot[k] += ctx->hostStats[b].prof[k];
        unsigned long long sum = 0; for (int k = 0; k < 16; ++k) sum += tot[k];
        if (sum) { std::fprintf(stderr, "[PT_PROFILE] k_shade wave-cycles by section:"); for (int k = 0; k < 16; ++k) if (k != 10 && k != 11) std::fprintf(stderr, " s%d=%.1f%%", k, 100.0*double(tot[k])/double(sum)); std::fprintf(stderr, " total=%llu\n", sum); }
        // per shading class, and how evenly the class's work is spread: per workgroup and per set of workgroups b, b + CUs, ... (one CU's, if
        // the dispatcher deals workgroups to the CUs in order)
        const size_t cus = size_t(ctx->prop.multiProcessorCount);
        for (int c = 0; c <= PT_NUM_CLASSES; ++c) {
            unsigned long long ct[16] = {0}, csum = 0;
            std::vector<double> perBlock(g, 0.0), perCu(std::min(cus, g), 0.0);
            for (size_t b = 0; b < g; ++b) for (int k = 0; k < 16; ++k) { ct[k] += ctx->hostStats[b].profCls[c][k]; perBlock[b] += double(ctx->hostStats[b].profCls[c][k]); }
            for (size_t b = 0; b < g; ++b) perCu[b % perCu.size()] += perBlock[b];
            for (int k = 0; k < 16; ++k) csum += ct[k];
            if (!csum) continue;
            csum -= ct[10];
            std::fprintf(stderr, "[PT_PROFILE] class %d: %llu wave turns, %.2f us per turn:", c, ct[10], ct[10] ? double(csum)*0.01/double(ct[10]) : 0.0);
            for (int k = 0; k < 16; ++k) if (k != 10 && k != 11) {
                unsigned long long ln = 0;
                for (size_t b = 0; b < g; ++b) ln += ctx->hostStats[b].profLanes[c][k];
                std::fprintf(stderr, " s%d=%.2fus(%.0f lanes)", k, ct[10] ? double(ct[k])*0.01/double(ct[10]) : 0.0, ct[k] ? double(ln)/double(ct[k]) : 0.0);
            }
            auto spread = [](std::vector<double> v, const char *what) {
                std::sort(v.begin(), v.end());
                double mean = 0.0; for (double x : v) mean += x; mean /= double(v.size());
                std::fprintf(stderr, " | %s min %.2f p50 %.2f p95 %.2f max %.2f of the mean", what, v.front()/mean, v[v.size()/2]/mean, v[v.size()*95/100]/mean, v.back()/mean);
            };
            spread(perBlock, "per workgroup");
            spread(perCu, "per CU");
            std::fprintf(stderr, " total=%llu\n", csum);
        }
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
    uint32_t cap = slotCap(ctx);
    if (isFlat(ctx) || ctx->haveInstances)
        cap = std::min<uint32_t>(cap, 16u*uint32_t(std::min(ctx->thrClosest, ctx->thrShadow))/64u*64u);   // OrderRegs
    perBlock = std::min<uint32_t>(cap, std::max<uint32_t>(64u, (perBlock + 63u)/64u*64u));
    const uint32_t slots = perBlock*grid;
    PathState &p = ctx->pool;
    // the walk arrays behind the A_* ones (suspended walks of the wide kernels, PathState::walk_base): only where they fit the 32-bit offsets
    uint32_t walkArrays = useWide(ctx) && !ctx->haveInstances && !ctx->poolRecords ? 4u + uint32_t(ctx->wideDepth + 1)/2u : 0u;
    if (ctx->poolSlots >= slots && ctx->poolGrid == grid && (ctx->poolWalkWanted == walkArrays || ctx->poolWalkArrays >= walkArrays)) {
        p.num_slots = slots;
        p.slots_per_block = perBlock;
        return TGHIP_OK;
    }
    ctx->poolWalkWanted = walkArrays;
    int rc = foldCounters(ctx);                  // the per-workgroup statistics live in the pool
    if (rc != TGHIP_OK) return rc;
    ctx->poolMem.release();
    ctx->poolSlots = 0;
#define POOL_ALLOC(field, n) do { std::remove_reference<decltype(*p.field)>::type *tmp_ = nullptr; \
        if ((rc = allocArray(ctx, ctx->poolMem, (n), &tmp_)) != TGHIP_OK) return rc; p.field = tmp_; } while (0)
    // arrays are skewed by an odd number of 256-byte units so that element i of different arrays does not map to
    // the same HBM channel (a power-of-two array stride made the kernels' speed depend on allocation luck)
    const uint64_t strideBytes = uint64_t(slots)*16u + uint64_t(ctx->poolPad);
    // (a group of eight arrays is addressed with a 32-bit offset from its own base, PathState::poolg)
    const uint64_t groupArrays = 1ull << PT_POOL_GROUP_SHIFT;
    if (strideBytes*groupArrays >= (1ull << 32)) { ctx->error = "path pool too large for 32-bit slot offsets within an array group"; return TGHIP_E_INVALID; }
    if (A_COUNT + walkArrays > groupArrays*PT_POOL_GROUPS) walkArrays = 0;
    ctx->poolWalkArrays = walkArrays;
    const uint64_t recordBytes = uint64_t(slots)*336u + 256u;   // the record layout: 128 + 128 + 80 bytes per slot
    const bool records = ctx->poolRecords && recordBytes < (1ull << 32);
    char *poolBase = nullptr;
    if ((rc = allocArray(ctx, ctx->poolMem, size_t(std::max<uint64_t>(strideBytes*(A_COUNT + walkArrays), records ? recordBytes : 0)), &poolBase)) != TGHIP_OK) return rc;
    for (uint32_t g = 0; g < PT_POOL_GROUPS; ++g)
        p.poolg[g] = poolBase + size_t(g)*size_t(groupArrays)*size_t(strideBytes);   // (groups past the last array are never addressed)
    p.walk_base = A_COUNT;
    p.stride = uint32_t(strideBytes);
    p.records = records ? 1u : 0u;
    p.rec_shadow = uint32_t(uint64_t(slots)*128u + 128u);
    p.rec_aux = uint32_t(uint64_t(slots)*256u + 256u);
    POOL_ALLOC(bm, size_t(slots/32)*Q_COUNT);
    p.bmStride = slots/32;
    POOL_ALLOC(ctl, grid); POOL_ALLOC(stats, grid); POOL_ALLOC(live, 4);
#undef POOL_ALLOC
    HIP_TRY(ctx, hipMemsetAsync(p.ctl, 0, sizeof(BlockCtl)*grid, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(p.stats, 0, sizeof(BlockStats)*grid, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(p.live, 0, 4*sizeof(uint32_t), ctx->stream));
    p.abort_flag = ctx->abortFlagDev;
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
    // Measured (profiles/README.md).  On one stream BVH scenes run best with every workgroup of every kernel resident at once
    // (4 per CU, workgroup size per kernel = that kernel's occupancy limit / 4: the fallback kernels still run so); flat-list scenes are
    // streaming-bound and prefer 8 small workgroups per CU that the dispatcher load-balances.
    // Scenes on the wide kernels run the loop as PARTS of the pool on streams of their own (runBatch), each kernel launched with its part
    // of the grid: with 8 workgroups per CU kernels of different parts share every CU, and the issue-bound walk of one part fills the
    // memory waits of the shading of another (materialtest 1280x720x256: 608 -> 657 Msamples/s against 4 per CU; workgroup sizes below
    // from the same sweep).  Instanced scenes with the dynamic-fetch closest-hit kernel: two parts, 8 per CU (instances10k: 141 -> 153
    // Msamples/s; with the static-fetch kernel parts lost: 127 against 114-120).
    const bool oneStream = ctx->streamsOpt == 1 || (ctx->streamsOpt == 0 && ctx->shortBatch);   // (shortBatch: tghip_render_pass)
    const bool pairedInst = !flat && ctx->haveInstances && !oneStream && !wideClosest(ctx) && ctx->dynamicFetch && ctx->instDynOpt && wideShadowRays(ctx) &&
                            !ctx->haveForward && !ctx->haveMeshLight;
    const bool paired = (!flat && !ctx->haveInstances && !oneStream && wideClosest(ctx) && wideShadowRays(ctx)) || pairedInst;
    ctx->blocksPerCu = ctx->blocksPerCuOpt > 0 ? ctx->blocksPerCuOpt : ((flat || paired) ? 8 : 4);
    if (flat && ctx->blocksPerCuOpt == 0) {
        ctx->thrClosest = ctx->thrShadow = ctx->thrShadeSimple = ctx->thrShadeComplex = 256;
    } else {
    const bool inst = ctx->haveInstances;
    const bool dyn = ctx->dynamicFetch && !inst;               // (the two-level dynamic-fetch kernel is chosen by instDynOpt below)
    const bool wide = useWide(ctx);
    const bool wideC = wideClosest(ctx), wideS = wideShadowRays(ctx);
    // (wide closest-hit kernel, materialtest 1280x720x256 / mesh1m, one MI355X: 128 / 192 / 256 / 320 threads = 505 / 433 / 394 / 469 us per
    // launch, but the shading launches behind it run 6 % faster after 192 than after 256: 579 / 571 / 552 Msamples/s for 192 / 256 / 320)
    ctx->thrClosest = wideC && inst ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest_wide<false, true, true>, 192, 3) : pickThreads(ctx, k_trace_closest_wide<false, false, true>, 192, 3))
                    : wideC ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest_wide<false, true>, 192, 3) : pickThreads(ctx, k_trace_closest_wide<false, false>, 192, 3))
                    : flat ? pickThreads(ctx, k_trace_closest<false, true>, 512, 1)
                    : (inst && ctx->dynamicFetch && ctx->instDynOpt && instWide(ctx)) ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest_instw<false, true>, 320, 4) : pickThreads(ctx, k_trace_closest_instw<false, false>, 320, 4))
                    : (inst && ctx->dynamicFetch && ctx->instDynOpt) ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest_inst<false, true>, 320, 2) : pickThreads(ctx, k_trace_closest_inst<false, false>, 320, 2))
                    : inst ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest<false, false, 1>, 512, 1) : pickThreads(ctx, k_trace_closest<false, false, 2>, 512, 1))
                    : dyn ? (ctx->haveSolids ? pickThreads(ctx, k_trace_closest_dyn<false, true>, 320, 2) : pickThreads(ctx, k_trace_closest_dyn<false, false>, 320, 2))   // 20 waves/CU measured best (profiles/README.md)
                                        : pickThreads(ctx, k_trace_closest<false, false>, 512, 1);
    if (wideS && inst && !ctx->haveForward && !ctx->haveMeshLight)
        ctx->thrShadow = ctx->instShadowFast ? (ctx->haveSolids ? pickThreads(ctx, k_trace_shadow_fast_inst<false, true>, 256, 3) : pickThreads(ctx, k_trace_shadow_fast_inst<false, false>, 256, 3))
                                             : (ctx->haveSolids ? pickThreads(ctx, k_trace_shadow_wide<false, true, true>, 256, 3) : pickThreads(ctx, k_trace_shadow_wide<false, false, true>, 256, 3));
    else if (wideS && !ctx->haveForward && !ctx->haveMeshLight)
        ctx->thrShadow = ctx->haveSolids ? pickThreads(ctx, k_trace_shadow_wide<false, true>, 256, 3) : pickThreads(ctx, k_trace_shadow_wide<false, false>, 256, 3);
    else if (!flat && !ctx->haveForward && !ctx->haveMeshLight && dyn)
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
    // (one workgroup size for the launches of classes 1 .. 3: that of the largest variant among them)
    if (ctx->haveMeshLight || inst || (ctx->classPresent[3] && (ctx->classMask[3] & ~MASK_PLASTIC) != 0))
                                                    ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_FULL, 2, 0>, 256, 0);
    else if (ctx->classPresent[3])                  ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_PLASTIC, 2, 0>, 256, 0);
    else if (ctx->classPresent[2])                  ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_GLASS, 2, 0>, 256, 0);
    else                                            ctx->thrShadeComplex = pickThreads(ctx, k_shade<MASK_COAT, COAT_WAVES, 0>, 256, 0);
    if (ctx->haveMedia) ctx->thrShadeSimple = ctx->thrShadeComplex = pickThreads(ctx, k_shade<BSDF_MASK_ALL, 2, 0>, 256, 0);
    if (paired && ctx->blocksPerCuOpt == 0) {
        // (closest / shadow / shade simple / shade complex, Msamples/s: 256/256/128/128 657, 192/256/128/128 634, 128/256/128/128 623,
        //  320/256/128/128 556, 256/320/128/128 549, 256/256/256/128 646, 256/256/128/64 623; 4 per CU with 192/256/192/128: 608)
        ctx->thrClosest = pairedInst ? 192 : 256;
        if (!ctx->haveForward && !ctx->haveMeshLight) ctx->thrShadow = 256;
        ctx->thrShadeSimple = ctx->thrShadeComplex = 128;
        // (round 5, the same sweep at today's kernels, three rounds A B C in one session, profiles/r5_sweep_shade_threads.jsonl: materialtest
        // 128/128 965, 192/192 967, 256/256 979 Msamples/s -- k_shade's 26 KB of LDS per workgroup then serve four waves instead of two --, but
        // mesh1m 604 -> 597, materialtest with a rough dielectric 453 -> 451 / 431, with a dielectric 480 -> 480: 256 threads only for what it
        // was measured to help, a small tree with the conductor family as its only other shading class)
        if (!pairedInst && ctx->numWideNodes <= 32768u && ctx->classPresent[1] && !ctx->classPresent[2] && !ctx->classPresent[3])
            ctx->thrShadeSimple = ctx->thrShadeComplex = 256;
    }
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
    {
        // (creation order matters: HIP deals its streams round-robin to a few hardware queues and streams on one hardware queue run their
        // kernels one after the other; the main stream and the part streams come first)
        StreamSet set;
        bool reused = false;
        {
            std::lock_guard<std::mutex> lock(g_streamMutex);
            std::vector<StreamSet> &pool = g_freeStreams[device_ordinal];
            if (!pool.empty()) { set = pool.back(); pool.pop_back(); reused = true; }
        }
        if (!reused) {
            // TGHIP_STREAM_PRIORITIES="p0,p1,..." (an experiment's hook, profiles/r6_ab_stream_priorities.txt): HIP stream priorities of the main stream and the part
            // streams, in creation order (lower = served first; the device's range is clamped by HIP); unset: plain streams
            int prio[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool withPrio = false;
            if (const char *env = std::getenv("TGHIP_STREAM_PRIORITIES")) {
                withPrio = true;
                int k = 0;
                for (const char *c = env; *c && k < 8; ) {
                    prio[k++] = int(std::strtol(c, const_cast<char **>(&c), 10));
                    while (*c == ',' || *c == ' ') ++c;
                }
            }
            auto create = [&](hipStream_t *st, int p) { return withPrio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, p) : hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
            if (e == hipSuccess) e = create(&set.main, prio[0]);
            for (int k = 0; k < 7; ++k)
                if (e == hipSuccess) e = create(&set.part[k], prio[k + 1]);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&set.abort, hipStreamNonBlocking);
        }
        ctx->stream = set.main;
        for (int k = 0; k < 7; ++k) ctx->partStream[k] = set.part[k];
        ctx->abortStream = set.abort;
    }
    for (int k = 0; k < 7; ++k)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evPart[k], hipEventDisableTiming);
    for (int k = 0; k < 8; ++k) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evFork[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evRot[k], hipEventDisableTiming);
        for (int a = 0; a < 2; ++a)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evJoin[k][a], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->evMain, hipEventDisableTiming);
    ctx->launchStream = ctx->stream;
    if (e == hipSuccess) e = hipEventCreate(&ctx->evA);
    if (e == hipSuccess) e = hipEventCreate(&ctx->evB);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ctx->hostLive), 2*sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ctx->abortFlagDev), 64);
    if (e == hipSuccess) e = hipMemset(ctx->abortFlagDev, 0, 64);
    if (e != hipSuccess) {
        {
            std::lock_guard<std::mutex> lock(g_errMutex);
            g_createError = std::string("tghip_create: ") + hipGetErrorString(e);
        }
        tghip_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

extern "C++" void tghipDestroyRankComm(void *comm);   // (defined with the RCCL loader below)
void tghip_destroy(tghip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // every stream idle before the first buffer goes (the part / class streams join the main one at the end of a pass, an aborted or
    // failed pass may have left them behind)
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 7; ++k) if (ctx->partStream[k]) (void)hipStreamSynchronize(ctx->partStream[k]);
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 2; ++a) if (ctx->classStream[k][a]) (void)hipStreamSynchronize(ctx->classStream[k][a]);
    if (ctx->abortStream) (void)hipStreamSynchronize(ctx->abortStream);
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
    if (ctx->dSamples) (void)hipFree(ctx->dSamples);
    if (ctx->dOwnedTiles) (void)hipFree(ctx->dOwnedTiles);
    if (ctx->rankComm) tghipDestroyRankComm(ctx->rankComm);
    if (ctx->redSum) (void)hipFree(ctx->redSum);
    if (ctx->redCount) (void)hipFree(ctx->redCount);
    if (ctx->partial) (void)hipFree(ctx->partial);
    if (ctx->hostLive) (void)hipHostFree(ctx->hostLive);
    if (ctx->abortFlagDev) (void)hipFree(ctx->abortFlagDev);
    if (ctx->evA) (void)hipEventDestroy(ctx->evA);
    if (ctx->evB) (void)hipEventDestroy(ctx->evB);
    for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
    for (int k = 0; k < 7; ++k) if (ctx->evPart[k]) (void)hipEventDestroy(ctx->evPart[k]);
    if (ctx->evMain) (void)hipEventDestroy(ctx->evMain);
    for (int k = 0; k < 8; ++k) {
        if (ctx->evFork[k]) (void)hipEventDestroy(ctx->evFork[k]);
        if (ctx->evRot[k]) (void)hipEventDestroy(ctx->evRot[k]);
        for (int a = 0; a < 2; ++a) {
            if (ctx->evJoin[k][a]) (void)hipEventDestroy(ctx->evJoin[k][a]);
            if (ctx->classStream[k][a]) (void)hipStreamDestroy(ctx->classStream[k][a]);
        }
    }
    {
        // the streams go back to the device's free list (idle: synchronised above)
        StreamSet set;
        set.main = ctx->stream; set.abort = ctx->abortStream;
        bool complete = set.main != nullptr && set.abort != nullptr;
        for (int k = 0; k < 7; ++k) { set.part[k] = ctx->partStream[k]; complete = complete && set.part[k] != nullptr; }
        if (complete) {
            std::lock_guard<std::mutex> lock(g_streamMutex);
            g_freeStreams[ctx->device].push_back(set);
        } else {                                 // (a context whose creation failed half way)
            if (set.main) (void)hipStreamDestroy(set.main);
            if (set.abort) (void)hipStreamDestroy(set.abort);
            for (int k = 0; k < 7; ++k) if (set.part[k]) (void)hipStreamDestroy(set.part[k]);
        }
    }
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
    else if (k == "slots_per_block") { ctx->slotsPerBlockOpt = int(std::min<long long>(std::max<long long>(value, 0), PT_MAX_SLOTS_PER_BLOCK))/64*64; ctx->poolMem.release(); ctx->poolSlots = 0; if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "chunk_samples") ctx->chunkSamples = int(std::min<long long>(std::max<long long>(value, 0), 1 << 20));
    else if (k == "check_interval") ctx->checkInterval = int(std::min<long long>(std::max<long long>(value, 0), 64));
    else if (k == "time_kernels") ctx->timeKernels = value != 0;
    else if (k == "streams") { ctx->streamsOpt = int(std::min<long long>(std::max<long long>(value, 0), 8)); if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "class_streams") {
        ctx->classStreamsOpt = value != 0;
        for (int kk = 0; kk < 8 && value != 0; ++kk)     // (created when the option is first switched on: an experiment's streams)
            for (int a = 0; a < 2; ++a)
                if (!ctx->classStream[kk][a]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->classStream[kk][a], hipStreamNonBlocking));
    }
    else if (k == "inst_simple") ctx->instSimpleOpt = value != 0;
    else if (k == "inst_dyn") { ctx->instDynOpt = value != 0; if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "shade_lds_pad") ctx->shadeLdsPad = int(std::min<long long>(std::max<long long>(value, 0), 120*1024));
    else if (k == "inst_shadow_fast") { ctx->instShadowFast = value != 0; if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "inst_wide") { ctx->instWideOpt = value != 0; if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "inst_phase_min") ctx->instPhaseMin = int(std::min<long long>(std::max<long long>(value, 1), 64));
    else if (k == "inst_refill_at") ctx->instRefillAt = int(std::min<long long>(std::max<long long>(value, 0), 63));
    else if (k == "grid_rounds") { ctx->gridRounds = int(std::min<long long>(std::max<long long>(value, 1), 8)); ctx->poolMem.release(); ctx->poolSlots = 0; }
    else if (k == "blocks_per_cu") { ctx->blocksPerCuOpt = int(std::min<long long>(std::max<long long>(value, 0), 8)); if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "leaf_batch_bvh2") ctx->leafBatchBvh2 = int(std::min<long long>(std::max<long long>(value, 0), 64));
    else if (k == "suspend_lanes") ctx->suspendLanes = int(std::min<long long>(std::max<long long>(value, 0), 64));
    else if (k == "suspend_turns") ctx->suspendTurns = int(std::min<long long>(std::max<long long>(value, 1), 1 << 20));   // (>= 1: every launch advances every walk)
    else if (k == "suspend_min_queue") ctx->suspendMinQueue = int(std::min<long long>(std::max<long long>(value, 0), 1 << 20));
    else if (k == "decouple") ctx->decoupleOpt = value != 0;
    else if (k == "inst_shadow_join") ctx->instShadowJoin = value != 0;
    else if (k == "fail_reduce") ctx->failReduce = value != 0;   // fault injection: tghip_reduce_framebuffers with this context as a rank fails (the hosts' fallbacks are tested with it)
    else if (k == "tail_kernel") ctx->tailOpt = value != 0;
    else if (k == "tail_family") ctx->tailFamilyOpt = value != 0;
    else if (k == "hoist_quad") { ctx->hoistOpt = value != 0; if (!ctx->hoistOpt) ctx->scene.hoisted_rec = -1; else ctx->scene.hoisted_rec = ctx->hoistedRecScene; }
    else if (k == "merge_miss") ctx->mergeMissOpt = value != 0;
    else if (k == "rotate_streams") ctx->rotateStreamsOpt = value != 0;
    else if (k == "fold_finish") ctx->foldFinishOpt = value != 0;
    else if (k == "finish_lean") ctx->finishLeanOpt = value != 0;
    else if (k == "top_tree") ctx->topTreeOpt = value != 0;
    else if (k == "lds_tables") ctx->tablesFit = ctx->tablesFitScene && value != 0;
    else if (k == "env_lds") ctx->scene.env_tex = value != 0 ? ctx->envTexScene : -1;
    else if (k == "media_lean") ctx->mediaLeanOpt = value != 0;
    else if (k == "tail_threshold") ctx->tailThreshold = value;
    else if (k == "lds_nodes") ctx->ldsNodesOpt = int(std::min<long long>(std::max<long long>(value, 0), 585));
    else if (k == "leaf_batch") ctx->leafBatch = int(std::min<long long>(std::max<long long>(value, 1), 64));
    else if (k == "fuse_flat") ctx->fuseFlatOpt = value != 0;
    else if (k == "run_to_completion") ctx->loopOpt = value != 0;
    else if (k == "pool_layout") {
#ifdef PT_POOL_RECORDS_RUNTIME
        ctx->poolRecords = value != 0; ctx->poolMem.release(); ctx->poolSlots = 0;
#else
        if (value != 0) { ctx->error = "pool_layout = 1 needs a build with -DPT_POOL_RECORDS_RUNTIME (the record layout was measured and not adopted)"; return TGHIP_E_UNSUPPORTED; }
#endif
    }
    else if (k == "pool_pad") { ctx->poolPad = std::max<long long>(value, 0)/16*16; ctx->poolMem.release(); ctx->poolSlots = 0; }
    else if (k == "wide_node_stride") {
        if (value != 128 && (PT_WIDE_HALF || value != 80)) { ctx->error = PT_WIDE_HALF ? "wide_node_stride is 128 (half-plane nodes)" : "wide_node_stride is 80 or 128"; return TGHIP_E_INVALID; }
        ctx->wideStride = int(value);
    }
    else if (k == "wide_closest") { ctx->wideClosestOpt = value < 0 ? -1 : value != 0; if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "wide_shadow") { ctx->wideShadowOpt = value < 0 ? -1 : value != 0; if (ctx->haveScene) chooseThreads(ctx); }
    else if (k == "wide_bvh") { ctx->wideOpt = value != 0; if (ctx->haveScene) chooseThreads(ctx); }
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
    // every index the shim (or a kernel) dereferences on the caller's word: refuse a malformed description instead of
    // reading out of bounds
    {
        auto bad = [&](const char *what) { ctx->error = std::string("malformed scene description: ") + what; return TGHIP_E_INVALID; };
        if ((sd->num_recs && (!sd->recs || !sd->tri_attrs)) || (sd->num_objects && !sd->objects) || (sd->num_bsdfs && !sd->bsdfs) ||
            (sd->num_textures && !sd->textures) || (sd->num_lights && !sd->lights) || (sd->num_infinite_lights && !sd->infinite_lights))
            return bad("a non-empty array is NULL");
        for (uint32_t i = 0; i < sd->num_lights; ++i)
            if (sd->lights[i] < 0 || uint32_t(sd->lights[i]) >= sd->num_objects) return bad("lights[] entry out of range");
        for (uint32_t i = 0; i < sd->num_infinite_lights; ++i)
            if (sd->infinite_lights[i] < 0 || uint32_t(sd->infinite_lights[i]) >= sd->num_objects) return bad("infinite_lights[] entry out of range");
        for (uint32_t i = 0; i < sd->num_recs; ++i) {
            const uint32_t kind = TGHIP_REC_KIND(sd->recs[i].meta);
            if (kind > TGHIP_REC_INSTANCE_SET) return bad("unknown primitive record kind");
            if (TGHIP_REC_OBJECT(sd->recs[i].meta) >= sd->num_objects) return bad("primitive record refers to an object out of range");
        }
        for (uint32_t i = 0; i < sd->num_objects; ++i) {
            const TgHipObject &o = sd->objects[i];
            if (o.bsdf < -1 || o.bsdf >= int32_t(sd->num_bsdfs)) return bad("object bsdf out of range");
            if (o.emission < -1 || o.emission >= int32_t(sd->num_textures)) return bad("object emission texture out of range");
            if (o.light < -1 || o.light >= int32_t(sd->num_lights)) return bad("object light index out of range");
            if (o.int_medium < -1 || o.ext_medium < -1 || o.int_medium >= int32_t(sd->num_media) || o.ext_medium >= int32_t(sd->num_media))
                return bad("primitive medium out of range");
        }
        for (uint32_t i = 0; i < sd->num_bsdfs; ++i) {
            const TgHipBsdf &b = sd->bsdfs[i];
            const int32_t nt = int32_t(sd->num_textures), nb = int32_t(sd->num_bsdfs);
            if (b.albedo < -1 || b.albedo >= nt || b.roughness < -1 || b.roughness >= nt || b.tex1 < -1 || b.tex1 >= nt)
                return bad("bsdf texture out of range");
            if (b.sub0 < -1 || b.sub0 >= nb || b.sub1 < -1 || b.sub1 >= nb) return bad("nested bsdf out of range");
        }
        if (sd->camera.medium < -1 || sd->camera.medium >= int32_t(sd->num_media)) return bad("camera medium out of range");
        // instanced scenes: bvhDepthOf (level 1) and the tight-box upload index recs[] / inst_tight_boxes[] by num_top_recs
        if (sd->num_instances != 0) {
            if (sd->num_top_recs == 0 || sd->num_top_recs > sd->num_recs) return bad("num_top_recs out of range for a scene with instances");
            if (!sd->inst_tight_boxes) return bad("scene with instances without inst_tight_boxes");
            if (sd->num_inst_prims && !sd->inst_prims) return bad("scene with instances without inst_prims");
        }
    }
    int depth = bvhDepthOf(sd, &ctx->bvhMasterDepth);
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
    ctx->wideDepth = 0;
    ctx->wideMasterDepth = 0;
    ctx->numWideNodes = 0;
    s.hoisted_rec = -1; ctx->hoistedRecScene = -1;
    if (sd->wide_nodes && sd->num_wide_nodes) {
        // the wide nodes and the primitive records share ONE allocation, so that a lane of the wide kernels addresses
        // "a node or a record" with one base pointer and one 32-bit offset
        const int wd = wideDepthOf(sd, &ctx->wideMasterDepth);
        if (wd < 0) { ctx->error = "malformed wide BVH"; return TGHIP_E_INVALID; }
        const size_t stride = size_t(ctx->wideStride);
        const size_t nodeBytes = (size_t(sd->num_wide_nodes)*stride + 127u) & ~size_t(127);
        const size_t recBytes = std::max<size_t>(sd->num_recs, 1)*sizeof(TgHipPrimRec);
        if (nodeBytes + recBytes < (1ull << 32)) {
            char *p = nullptr;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&p), nodeBytes + recBytes));
            ctx->sceneMem.allocs.push_back(p);
#if PT_WIDE_HALF
            static_assert(sizeof(TgHipWideNode) == 80 && offsetof(TgHipWideNode, qlo) == 32 && offsetof(TgHipWideNode, qhi) == 56, "TgHipWideNode layout");
            // the device's node: the ABI's header, then the 48 child planes as halfs, one 16-byte row per axis and side (pt_kernels.h)
            {
                auto half = [](uint32_t q) -> uint16_t {          // the integer q <= 255 as an IEEE half (exact)
                    if (q == 0u) return uint16_t(0);
                    const uint32_t e = 31u - uint32_t(__builtin_clz(q));
                    return uint16_t(((e + 15u) << 10) | ((q << (10u - e)) & 0x3FFu));
                };
                std::vector<unsigned char> dev(size_t(sd->num_wide_nodes)*stride, 0);
                for (uint32_t i = 0; i < sd->num_wide_nodes; ++i) {
                    const TgHipWideNode &n = sd->wide_nodes[i];
                    unsigned char *d = dev.data() + size_t(i)*stride;
                    std::memcpy(d, &n, 32);
                    uint16_t *rows = reinterpret_cast<uint16_t *>(d + 32);
                    for (int a = 0; a < 3; ++a)
                        for (int sl = 0; sl < 8; ++sl) {
                            rows[a*8 + sl] = half(n.qlo[a][sl]);
                            rows[24 + a*8 + sl] = half(n.qhi[a][sl]);
                        }
                }
                HIP_TRY(ctx, hipMemcpy(p, dev.data(), dev.size(), hipMemcpyHostToDevice));
            }
#else
            HIP_TRY(ctx, hipMemcpy2DAsync(p, stride, sd->wide_nodes, sizeof(TgHipWideNode), sizeof(TgHipWideNode), sd->num_wide_nodes, hipMemcpyHostToDevice, ctx->stream));
#endif
            HIP_TRY(ctx, hipMemcpyAsync(p + nodeBytes, sd->recs, size_t(sd->num_recs)*sizeof(TgHipPrimRec), hipMemcpyHostToDevice, ctx->stream));
            // The scene's one quad, hoisted out of the decoupled walks (pt_scene.h: DeviceScene::hoisted_rec): a single-level scene of triangles and
            // exactly ONE quad whose wide nodes leave `reserved` zero.  The wide node that holds the quad as a leaf record gets the record's bit of
            // leaf_valid in its `reserved` word -- on the device copy only; the sequential walks (tghip_trace_rays, "decouple" = 0) do not read it.
            s.hoisted_rec = -1; ctx->hoistedRecScene = -1;
            if (!PT_WIDE_HALF && sd->num_instances == 0) {
                int64_t quad = -1;
                bool ok = true;
                for (uint32_t i = 0; i < sd->num_recs && ok; ++i) {
                    const uint32_t kind = TGHIP_REC_KIND(sd->recs[i].meta);
                    if (kind == TGHIP_REC_QUAD) { ok = quad < 0; quad = i; }
                    else if (kind != TGHIP_REC_TRIANGLE) ok = false;
                }
                int64_t node = -1;
                uint32_t bit = 0;
                for (uint32_t i = 0; i < sd->num_wide_nodes && ok && quad >= 0; ++i) {
                    const TgHipWideNode &n = sd->wide_nodes[i];
                    if (n.reserved != 0u) ok = false;
                    uint32_t rank = 0;
                    for (uint32_t b = 0; b < 32u; ++b)
                        if ((n.leaf_valid >> b) & 1u) {
                            if (int64_t(n.rec_base) + rank == quad) { ok = ok && node < 0; node = i; bit = b; }
                            ++rank;
                        }
                }
                if (ok && quad >= 0 && node >= 0) {
                    const uint32_t word = 1u << bit;
                    // (on ctx->stream, behind the 2-D copy of the nodes queued above: that stream is non-blocking, so a copy on the null stream would
                    // not wait for it and could be overwritten by it -- the quad would then be tested before AND inside every walk; `word` is on
                    // the stack, so the copy is waited for here)
                    HIP_TRY(ctx, hipMemcpyAsync(p + size_t(node)*stride + offsetof(TgHipWideNode, reserved), &word, sizeof(word), hipMemcpyHostToDevice, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    s.hoisted_rec = ctx->hoistOpt ? int32_t(quad) : -1;
                    ctx->hoistedRecScene = int32_t(quad);
                }
            }
            s.wide = reinterpret_cast<const float4 *>(p);
            s.recs_offset = uint32_t(nodeBytes);
            s.wide_stride = uint32_t(stride);
            dr = reinterpret_cast<const TgHipPrimRec *>(p + nodeBytes);
            ctx->wideDepth = wd;
            ctx->numWideNodes = sd->num_wide_nodes;
        }
    }
    if (!ctx->wideDepth && (rc = uploadArray(ctx, ctx->sceneMem, sd->recs, sd->num_recs, &dr)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->tri_attrs, sd->num_recs, &da)) != TGHIP_OK) return rc;
    s.nodes = reinterpret_cast<const float4 *>(dn);
    s.recs = reinterpret_cast<const float4 *>(dr);
    s.tri_attrs = reinterpret_cast<const float4 *>(da);
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->objects, sd->num_objects, &s.objects)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->lights, sd->num_lights, &s.lights)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->infinite_lights, sd->num_infinite_lights, &s.infinite_lights)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->bsdfs, sd->num_bsdfs, &s.bsdfs)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->textures, sd->num_textures, &s.textures)) != TGHIP_OK) return rc;
    // (16 bytes of room behind the texels: bitmapTexel reads three floats of a one-channel texture's texel too, pt_scene.h)
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->texels, sd->num_texel_floats, &s.texels, 16)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->dist, sd->num_dist_floats, &s.dist)) != TGHIP_OK) return rc;
    if ((rc = uploadArray(ctx, ctx->sceneMem, sd->light_tris, sd->num_light_tri_floats, &s.light_tris)) != TGHIP_OK) return rc;
    ctx->haveMeshLight = false;
    ctx->haveInstances = sd->num_instances > 0;
    ctx->haveMedia = sd->num_media > 0;
    ctx->haveCylinder = false;
    for (uint32_t i = 0; i < sd->num_objects; ++i)
        if (sd->objects[i].type == TGHIP_OBJ_CYLINDER) ctx->haveCylinder = true;
    for (uint32_t i = 0; i < sd->num_bsdfs; ++i) {
        if (sd->bsdfs[i].bump1 < 0 || uint32_t(sd->bsdfs[i].bump1) > sd->num_textures) { ctx->error = "bsdf bump map index out of range"; return TGHIP_E_INVALID; }
        if (sd->bsdfs[i].bump1 > 0) ctx->haveCylinder = true;      // (shaded by the same one variant)
    }
    if (ctx->haveMedia) {
        if (!sd->media || sd->num_media > PT_MAX_MEDIA) { ctx->error = "more than 126 media are not supported"; return TGHIP_E_UNSUPPORTED; }
        if (sd->num_objects >= (1u << 16)) { ctx->error = "media scenes support at most 65535 primitives"; return TGHIP_E_UNSUPPORTED; }
        if (sd->camera.medium >= int32_t(sd->num_media)) { ctx->error = "camera medium out of range"; return TGHIP_E_INVALID; }
        for (uint32_t i = 0; i < sd->num_objects; ++i)
            if (sd->objects[i].int_medium >= int32_t(sd->num_media) || sd->objects[i].ext_medium >= int32_t(sd->num_media)) {
                ctx->error = "primitive medium out of range";
                return TGHIP_E_INVALID;
            }
        for (uint32_t i = 0; i < sd->num_media; ++i) {
            if (sd->media[i].phase_type < TGHIP_PHASE_ISOTROPIC || sd->media[i].phase_type > TGHIP_PHASE_RAYLEIGH) {
                ctx->error = "unknown phase function";
                return TGHIP_E_UNSUPPORTED;
            }
            if (sd->media[i].medium_type < TGHIP_MEDIUM_HOMOGENEOUS || sd->media[i].medium_type > TGHIP_MEDIUM_ATMOSPHERE) {
                ctx->error = "unknown medium type"; return TGHIP_E_UNSUPPORTED;
            }
            if (sd->media[i].medium_type != TGHIP_MEDIUM_HOMOGENEOUS && sd->media[i].trans_type != TGHIP_TRANS_EXPONENTIAL) {
                ctx->error = "an exponential or atmospheric medium with a non-exponential transmittance is not supported"; return TGHIP_E_UNSUPPORTED;
            }
            if (sd->media[i].medium_type == TGHIP_MEDIUM_ATMOSPHERE && !(sd->media[i].falloff_scale > 0.0f && sd->media[i].falloff_dir[0] > 0.0f)) {
                ctx->error = "an atmospheric medium needs a positive falloff scale and radius"; return TGHIP_E_INVALID;
            }
            if (sd->media[i].trans_type < TGHIP_TRANS_EXPONENTIAL || sd->media[i].trans_type > TGHIP_TRANS_INTERPOLATED) {
                ctx->error = "unknown transmittance";
                return TGHIP_E_UNSUPPORTED;
            }
            if (sd->media[i].trans_type == TGHIP_TRANS_INTERPOLATED &&
                (i + 2 >= sd->num_media || sd->media[i + 1].trans_type == TGHIP_TRANS_INTERPOLATED || sd->media[i + 2].trans_type == TGHIP_TRANS_INTERPOLATED)) {
                ctx->error = "an interpolated transmittance needs its two (non-interpolated) operands in the media entries behind it";
                return TGHIP_E_INVALID;
            }
        }
    }
    ctx->thinlens = sd->camera.type == TGHIP_CAMERA_THINLENS;
    // the equirectangular camera's rays are written by a launch of their own in front of every closest-hit launch (k_camera_rays): the kernels that
    // generate a camera ray and trace it in one go -- the folded finish, the flat lists' fused launches, k_tail -- are not used for such scenes
    ctx->cameraFix = sd->camera.type == TGHIP_CAMERA_EQUIRECTANGULAR || sd->camera.type == TGHIP_CAMERA_CUBEMAP;
    if (sd->camera.type == TGHIP_CAMERA_CUBEMAP && (sd->camera.blade_count < 0 || sd->camera.blade_count > 3)) { ctx->error = "unknown cubemap projection mode"; return TGHIP_E_INVALID; }
    if (sd->camera.type < TGHIP_CAMERA_PINHOLE || sd->camera.type > TGHIP_CAMERA_CUBEMAP) { ctx->error = "unknown camera type"; return TGHIP_E_UNSUPPORTED; }
    if (sd->camera.type == TGHIP_CAMERA_THINLENS && sd->camera.aperture_type == TGHIP_APERTURE_BITMAP) {
        // the aperture's Distribution2D: marginalPdf[h] marginalCdf[h + 1] pdf[w h] cdf[(w + 1) h] inside dist[]
        const uint64_t aw = uint64_t(std::max(sd->camera.aperture_w, 0)), ah = uint64_t(std::max(sd->camera.aperture_h, 0));
        if (aw == 0 || ah == 0 || !sd->dist || uint64_t(sd->camera.aperture_dist) + ah + ah + 1 + aw*ah + (aw + 1)*ah > sd->num_dist_floats) {
            ctx->error = "the bitmap aperture's distribution lies outside dist[]";
            return TGHIP_E_INVALID;
        }
        // a table that cannot be inverted (an all-black aperture: 0/0 in the marginal CDF) would send every lens sample outside it
        const float *mcdf = sd->dist + sd->camera.aperture_dist + ah;
        bool usable = mcdf[0] == 0.0f && mcdf[ah] > 0.0f;
        for (uint64_t i = 0; usable && i < ah; ++i)
            usable = std::isfinite(mcdf[i + 1]) && mcdf[i + 1] >= mcdf[i];
        if (!usable) {
            ctx->error = "the bitmap aperture's distribution is not a CDF (zero total weight or non-finite entries)";
            return TGHIP_E_INVALID;
        }
    } else if (sd->camera.type == TGHIP_CAMERA_THINLENS && sd->camera.aperture_type != TGHIP_APERTURE_DISK && sd->camera.aperture_type != TGHIP_APERTURE_BLADE) {
        ctx->error = "unknown aperture type";
        return TGHIP_E_UNSUPPORTED;
    }
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
        // the conditional tables of those bitmaps once more as interleaved (cdf, pdf) pairs (pt_scene.h: upperBoundGuidedPairs)
        std::vector<float2> rows;
        std::vector<int32_t> texRows(std::max<uint32_t>(sd->num_textures, 1u), -1);
        for (uint32_t i = 0; i < sd->num_textures; ++i) {
            const TgHipTexture &t = sd->textures[i];
            if (texGuide[i] < 0 || rows.size() + size_t(t.w + 1)*size_t(t.h) >= (1u << 28))
                continue;
            texRows[i] = int32_t(rows.size());
            const float *pdf = sd->dist + t.dist_offset + t.h + (t.h + 1);
            const float *cdf = pdf + size_t(t.w)*t.h;
            for (int y = 0; y < t.h; ++y)
                for (int x = 0; x <= t.w; ++x)
                    rows.push_back(make_float2(cdf[size_t(y)*(t.w + 1) + x], x < t.w ? pdf[size_t(y)*t.w + x] : 0.0f));
        }
        if (rows.empty()) rows.push_back(make_float2(0.0f, 0.0f));
        if ((rc = uploadArray(ctx, ctx->sceneMem, rows.data(), rows.size(), &s.rows)) != TGHIP_OK) return rc;
        if ((rc = uploadArray(ctx, ctx->sceneMem, texRows.data(), texRows.size(), &s.tex_rows)) != TGHIP_OK) return rc;
        // the marginal tables of the first sampled environment map, for the shading kernels' LDS copy (stageSceneTables)
        s.env_tex = -1; s.env_h = 0; s.env_marginal = nullptr; s.env_guide = nullptr;
        std::vector<uint16_t> envGuide;
        for (uint32_t li = 0; li < sd->num_lights && s.env_tex < 0; ++li) {
            const TgHipObject &o = sd->objects[sd->lights[li]];
            if (o.type != TGHIP_OBJ_INFINITE_SPHERE || o.emission < 0 || texGuide[size_t(o.emission)] < 0) continue;
            const TgHipTexture &t = sd->textures[o.emission];
            s.env_tex = o.emission;
            s.env_h = t.h;
            s.env_marginal = s.dist + t.dist_offset;          // mpdf[h] mcdf[h + 1] (device pointer arithmetic only)
            envGuide.assign(guide.begin() + texGuide[size_t(o.emission)], guide.begin() + texGuide[size_t(o.emission)] + PT_GUIDE_MARGINAL + 1);
            envGuide.push_back(0);                            // padded to whole 32-bit words
        }
        // do the small tables fit the shading workgroups' LDS copy (pt_kernels.h: stageSceneTables)?  Without the environment map's marginal
        // tables they must, or the scene shades with the GLOBAL_TABLES variant; the marginal tables come along only when there is room
        ctx->tablesFit = sceneTableLayout(sd->num_objects, sd->num_bsdfs, sd->num_textures, sd->num_lights, sd->num_infinite_lights, 0).total <= PT_LDS_TABLE_BYTES;
        if (s.env_tex >= 0 && sceneTableLayout(sd->num_objects, sd->num_bsdfs, sd->num_textures, sd->num_lights, sd->num_infinite_lights, s.env_h).total > PT_LDS_TABLE_BYTES) {
            s.env_tex = -1; s.env_h = 0; s.env_marginal = nullptr;      // (sampled through the texture's own tables in global memory, like any other bitmap)
            envGuide.clear();
        }
        ctx->tablesFitScene = ctx->tablesFit;
        ctx->envTexScene = s.env_tex;
        if (envGuide.empty()) envGuide.assign(2, 0);
        if ((rc = uploadArray(ctx, ctx->sceneMem, envGuide.data(), envGuide.size(), &s.env_guide)) != TGHIP_OK) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the vectors go out of scope
    }
    // shading classes ("sort by material", pt_kernels.h: PT_NUM_CLASSES): 0 = BSDFs made of lambert / null only, 1 = conductor family, 2 = dielectric family, 3 = the rest
    {
        std::vector<uint32_t> typeMask(sd->num_bsdfs, 0u);
        std::vector<uint8_t> recClass(std::max<uint32_t>(sd->num_recs, 1u), 0);
        ctx->haveComplex = false; ctx->complexMask = 0; ctx->haveForward = false; ctx->haveSolids = false;
        for (int c = 0; c < PT_NUM_CLASSES; ++c) { ctx->classPresent[c] = false; ctx->classMask[c] = 0; }
        for (uint32_t i = 0; i < sd->num_bsdfs; ++i) {
            typeMask[i] = bsdfTypeMask(sd, int(i), 0);
            if (sd->bsdfs[i].lobes & TGHIP_LOBE_FORWARD) ctx->haveForward = true;
        }
        for (uint32_t i = 0; i < sd->num_recs; ++i) {
            uint32_t meta = sd->recs[i].meta;
            if (TGHIP_REC_KIND(meta) == TGHIP_REC_INSTANCE || TGHIP_REC_KIND(meta) == TGHIP_REC_INSTANCE_SET)
                continue;                    // never a hit record itself: hits are the master's triangles
            if (TGHIP_REC_KIND(meta) > TGHIP_REC_CYLINDER) { ctx->error = "unknown primitive record kind"; return TGHIP_E_INVALID; }
            if (TGHIP_REC_KIND(meta) != TGHIP_REC_TRIANGLE && TGHIP_REC_KIND(meta) != TGHIP_REC_QUAD) ctx->haveSolids = true;
            int bi = TGHIP_REC_KIND(meta) == TGHIP_REC_TRIANGLE ? sd->tri_attrs[i].bsdf : sd->objects[TGHIP_REC_OBJECT(meta)].bsdf;
            if (bi < 0 || uint32_t(bi) >= sd->num_bsdfs) { ctx->error = "primitive record without a valid bsdf"; return TGHIP_E_INVALID; }
            // the smallest family that covers every type inside the material (nested ones included); forward lobes -> "everything else"
            const uint32_t tm = typeMask[size_t(bi)];
            const bool fwd = (sd->bsdfs[bi].lobes & TGHIP_LOBE_FORWARD) != 0;
            const int c = (!fwd && (tm & ~MASK_SIMPLE) == 0) ? 0 : (!fwd && (tm & ~MASK_COAT) == 0) ? 1 : (!fwd && (tm & ~MASK_GLASS) == 0) ? 2 : 3;
            recClass[i] = uint8_t(c);
            ctx->classPresent[c] = true;
            ctx->classMask[c] |= tm;
            if (c != 0) { ctx->haveComplex = true; ctx->complexMask |= tm; }
        }
        {
            uint32_t allTypes = 0;
            for (int c = 0; c < PT_NUM_CLASSES; ++c) allTypes |= ctx->classMask[c];
            ctx->mediaSimple = ctx->haveMedia && !ctx->haveInstances && (allTypes & ~MASK_MEDIA & 0x7FFFFu) == 0;   // (bits 0 .. 18: the BSDF types)
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
    s.inst_prims = nullptr; s.inst_leaf_boxes = nullptr; s.inst_tight_boxes = nullptr;
    if (sd->num_instances) {
        if ((rc = uploadArray(ctx, ctx->sceneMem, sd->inst_prims, size_t(sd->num_inst_prims), &s.inst_prims)) != TGHIP_OK) return rc;
        const float4 *boxes = nullptr;
        if ((rc = uploadArray(ctx, ctx->sceneMem, reinterpret_cast<const float4 *>(sd->inst_leaf_boxes), size_t(sd->num_inst_prims)*2, &boxes)) != TGHIP_OK) return rc;
        s.inst_leaf_boxes = boxes;
        if (!sd->inst_tight_boxes) { ctx->error = "scene with instances without inst_tight_boxes"; return TGHIP_E_INVALID; }
        const float4 *tight = nullptr;
        if ((rc = uploadArray(ctx, ctx->sceneMem, reinterpret_cast<const float4 *>(sd->inst_tight_boxes), size_t(sd->num_top_recs)*2, &tight)) != TGHIP_OK) return rc;
        s.inst_tight_boxes = tight;
    }
    s.top_nodes = nullptr;
    s.flat_boxes = nullptr;
    if (sd->top_nodes && sd->num_top_nodes && ctx->topTreeOpt) {
        // the reference's top-level Embree tree (TgHipTopNode): flat lists only, every record exactly one leaf, children behind their parents
        // (preorder: no cycles), no deeper than the walk's stack allows (pt_kernels.h: flatOrderedWalk)
        const uint32_t nn = sd->num_top_nodes;
        bool ok = sd->num_recs >= 2 && sd->num_recs <= TGHIP_FLAT_MAX_RECS && !sd->num_instances && nn < sd->num_recs;
        std::vector<float4> boxes(size_t(sd->num_recs)*2);
        std::vector<int> leafOf(sd->num_recs, 0), depth(nn, 0), parents(nn, 0);
        if (ok) depth[0] = 1;
        for (uint32_t n = 0; n < nn && ok; ++n) {
            ok = depth[n] >= 1 && depth[n] <= TGHIP_TOP_MAX_DEPTH && (n == 0 || parents[n] == 1);
            for (int i = 0; i < 4 && ok; ++i) {
                const int32_t c = sd->top_nodes[n].child[i];
                if (c == TGHIP_TOP_EMPTY) continue;
                if (c >= 0) {
                    ok = uint32_t(c) > n && uint32_t(c) < nn;
                    if (ok) { depth[c] = depth[n] + 1; parents[c]++; }
                } else {
                    const uint32_t r = uint32_t(~c);
                    ok = r < sd->num_recs && leafOf[r]++ == 0;
                    if (ok) {
                        const TgHipTopNode &t = sd->top_nodes[n];
                        boxes[2*r] = make_float4(t.lower[i][0], t.lower[i][1], t.lower[i][2], 0.0f);
                        boxes[2*r + 1] = make_float4(t.upper[i][0], t.upper[i][1], t.upper[i][2], 0.0f);
                    }
                }
            }
        }
        for (uint32_t r = 0; r < sd->num_recs && ok; ++r) {
            const uint32_t kind = TGHIP_REC_KIND(sd->recs[r].meta);
            ok = leafOf[r] == 1 && (kind == TGHIP_REC_QUAD || kind == TGHIP_REC_CUBE || kind == TGHIP_REC_SPHERE || kind == TGHIP_REC_DISK || kind == TGHIP_REC_CYLINDER);
        }
        if (!ok) { ctx->error = "top_nodes: not the tree of a flat list (every record one leaf, preorder, depth <= TGHIP_TOP_MAX_DEPTH)"; return TGHIP_E_INVALID; }
        static_assert(sizeof(TgHipTopNode) == 28*sizeof(float), "TgHipTopNode layout");
        if ((rc = uploadArray(ctx, ctx->sceneMem, reinterpret_cast<const float *>(sd->top_nodes), size_t(nn)*28, &s.top_nodes)) != TGHIP_OK) return rc;
        if ((rc = uploadArray(ctx, ctx->sceneMem, boxes.data(), boxes.size(), &s.flat_boxes)) != TGHIP_OK) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // boxes goes out of scope
    }
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
    if (!ctx->tablesFit) {
        // the scene's small tables do not fit the shading workgroups' LDS copy (pt_kernels.h: stageSceneTables): every class shades with the
        // all-features variant that reads them from global memory (runBatch keeps such scenes off the fused and the tail launches)
        hipLaunchKernelGGL((k_shade<BSDF_MASK_ALL, 2, 0, true>), dim3(grid), dim3(ctx->thrShadeAll), 0, ctx->launchStream, ctx->scene, st, pp, cls);
        return;
    }
    hipLaunchKernelGGL((k_shade<M, ((B == MASK_SIMPLE || B == MASK_SIMPLE_INST) ? SIMPLE_WAVES : B == MASK_LEAN ? LEAN_WAVES : B == MASK_COAT ? COAT_WAVES : 2), FUSE>), dim3(grid),
                       dim3((M == BSDF_MASK_ALL || M == MASK_MEDIA) ? ctx->thrShadeAll : (cls >= 1 && cls < PT_NUM_CLASSES) ? ctx->thrShadeComplex : ctx->thrShadeSimple), size_t(ctx->shadeLdsPad), ctx->launchStream, ctx->scene, st, pp, cls);
}
// TGHIP_PASS_SOBOL / TGHIP_PASS_RECORDS passes run the FEAT_QMC twin of the variant the scene would use anyway
template<uint32_t M, int FUSE = 0>
static void launchShade(tghip_ctx *ctx, int grid, const PathState &st, const PassParams &pp, int cls)
{
    if (pp.flags) launchShadeVariant<M | FEAT_QMC, FUSE>(ctx, grid, st, pp, cls);
    else          launchShadeVariant<M, FUSE>(ctx, grid, st, pp, cls);
}

// the launch of shading class `cls` (1 .. 3) of a scene without media / auxiliary outputs / cylinders / mesh emitters: the smallest variant
// that covers the BSDF types of the class's materials (pt_kernels.h: PT_NUM_CLASSES)
template<int FUSE>
static void launchComplexClass(tghip_ctx *ctx, int grid, const PathState &st, const PassParams &pp, int cls)
{
    const bool plasticOnly = (ctx->classMask[3] & ~MASK_PLASTIC) == 0;
    if constexpr (FUSE == 0) {
        if (ctx->haveInstances) {
            if (cls == 1)                     launchShade<MASK_COAT_INST>(ctx, grid, st, pp, cls);
            else if (cls == 2)                launchShade<MASK_GLASS_INST>(ctx, grid, st, pp, cls);
            else if (plasticOnly)             launchShade<MASK_PLASTIC_INST>(ctx, grid, st, pp, cls);
            else                              launchShade<MASK_FULL>(ctx, grid, st, pp, cls);
            return;
        }
    }
    if (cls == 1)         launchShade<MASK_COAT, FUSE>(ctx, grid, st, pp, cls);
    else if (cls == 2)    launchShade<MASK_GLASS, FUSE>(ctx, grid, st, pp, cls);
    else if (plasticOnly) launchShade<MASK_PLASTIC, FUSE>(ctx, grid, st, pp, cls);
    else                  launchShade<MASK_FULL, FUSE>(ctx, grid, st, pp, cls);
}

// true when the shadow step needs its second half, k_finish (the dynamic-fetch kernel does not regenerate paths itself)
template<bool COUNT>
static bool launchShadow(tghip_ctx *ctx, int grid, const PathState &st, const PassParams &pp, uint32_t iterTag)
{
    const size_t ldsBytes = traceLdsBytes(ctx, ctx->thrShadow);
    const bool flat = isFlat(ctx);
    const bool closestWalk = ctx->haveForward || ctx->haveMeshLight;   // shadow rays are closest-hit walks, not any-hit queries
    const bool wideShadow = wideShadowRays(ctx) && !closestWalk && !ctx->auxPass;
    if (ctx->haveInstances && !wideShadow) {
#define SHADOW_INST(FWD, I) hipLaunchKernelGGL((k_trace_shadow<COUNT, FWD, false, I>), dim3(grid), dim3(ctx->thrShadow), ldsBytes, ctx->launchStream, ctx->scene, st, pp, iterTag)
        if (closestWalk) { if (ctx->haveSolids) SHADOW_INST(true, 1); else SHADOW_INST(true, 2); }
        else             { if (ctx->haveSolids) SHADOW_INST(false, 1); else SHADOW_INST(false, 2); }
#undef SHADOW_INST
        return false;
    }
    if (wideShadow) {
        const size_t lds = wideLdsBytes(ctx, ctx->thrShadow);
#define SHADOW_WIDE(S, I) hipLaunchKernelGGL((k_trace_shadow_wide<COUNT, S, I>), dim3(grid), dim3(ctx->thrShadow), lds, ctx->launchStream, ctx->scene, st, pp, iterTag)
        if (ctx->haveInstances) {
            // (rounds 2 and 3 launched the counting variant here: the non-counting one lost occluders inside instances.  The cause is the
            // loop latch the backend generates for the turn without PT_TURN_JOIN, pt_wavefront.h; "inst_shadow_join" = 0 launches that
            // variant for tools/repro_latch_miscompile.py)
            if (!ctx->instShadowJoin && !ctx->haveSolids && !COUNT)
                hipLaunchKernelGGL((k_trace_shadow_wide<false, false, true, false>), dim3(grid), dim3(ctx->thrShadow), lds, ctx->launchStream, ctx->scene, st, pp, iterTag);
            else if (ctx->instShadowFast) {      // round 6: the two-level walk inside the fast kernel's slot handling
                if (ctx->haveSolids) hipLaunchKernelGGL((k_trace_shadow_fast_inst<COUNT, true>), dim3(grid), dim3(ctx->thrShadow), lds, ctx->launchStream, ctx->scene, st, pp, iterTag);
                else                 hipLaunchKernelGGL((k_trace_shadow_fast_inst<COUNT, false>), dim3(grid), dim3(ctx->thrShadow), lds, ctx->launchStream, ctx->scene, st, pp, iterTag);
            }
            else if (ctx->haveSolids) SHADOW_WIDE(true, true); else SHADOW_WIDE(false, true);
        }
        else if (ctx->decoupleOpt) {
#define SHADOW_WIDE_D(S) hipLaunchKernelGGL((k_trace_shadow_fast<COUNT, S>), dim3(grid), dim3(ctx->thrShadow), lds, ctx->launchStream, ctx->scene, st, pp, iterTag)
            if (ctx->haveSolids) SHADOW_WIDE_D(true); else SHADOW_WIDE_D(false);
#undef SHADOW_WIDE_D
        }
        else                    { if (ctx->haveSolids) SHADOW_WIDE(true, false); else SHADOW_WIDE(false, false); }
#undef SHADOW_WIDE
        return true;
    }
    if (!flat && !closestWalk && ctx->dynamicFetch && !ctx->auxPass) {   // (the dynamic-fetch kernel does not report transmittances)
        if (ctx->haveSolids) hipLaunchKernelGGL((k_trace_shadow_dyn<COUNT, true>), dim3(grid), dim3(ctx->thrShadow), dynLdsBytes(ctx, ctx->thrShadow), ctx->launchStream,
                                                ctx->scene, st, pp, iterTag);
        else                 hipLaunchKernelGGL((k_trace_shadow_dyn<COUNT, false>), dim3(grid), dim3(ctx->thrShadow), dynLdsBytes(ctx, ctx->thrShadow), ctx->launchStream,
                                                ctx->scene, st, pp, iterTag);
        return true;
    }
#define SHADOW_LAUNCH(FWD, FLAT) hipLaunchKernelGGL((k_trace_shadow<COUNT, FWD, FLAT>), dim3(grid), dim3(ctx->thrShadow), ldsBytes, ctx->launchStream, ctx->scene, st, pp, iterTag)
    if (closestWalk) { if (flat) SHADOW_LAUNCH(true, true); else SHADOW_LAUNCH(true, false); }
    else                  { if (flat) SHADOW_LAUNCH(false, true); else SHADOW_LAUNCH(false, false); }
#undef SHADOW_LAUNCH
    return false;
}

} // extern "C++"

// Runs the wavefront loop for one batch of work items until every slot is drained.
static int runBatch(tghip_ctx *ctx, const PassParams &pp)
{
    if (ctx->abortRequested.load(std::memory_order_acquire))
        return TGHIP_E_ABORTED;                  // aborted before this batch started: nothing to drain
    PathState st = ctx->pool;
    st.partial = ctx->partial;
    st.abort_flag = ctx->abortFlagDev;
    st.leaf_batch = uint32_t(ctx->leafBatch);
    st.inst_tree_depth = uint32_t(instTreeDepth(ctx));
    st.inst_phase_min = uint32_t(ctx->instPhaseMin);
    st.inst_refill_at = uint32_t(ctx->instRefillAt);
    st.nee_factors = (ctx->haveForward || ctx->haveMeshLight) ? 1u : 0u;   // launchShadow: the closest-hit shadow walk, k_trace_shadow<., FORWARD>
    st.lds_nodes = ldsNodeCount(ctx);
    st.wide_depth = uint32_t(std::max(ctx->wideDepth, 1));
    st.suspend_lanes = ctx->poolWalkArrays ? uint32_t(ctx->suspendLanes) : 0u;
    st.suspend_turns = uint32_t(std::max(ctx->suspendTurns, 1));
    st.suspend_min_queue = uint32_t(ctx->suspendMinQueue);
    st.leaf_batch_bvh2 = uint32_t(ctx->leafBatchBvh2 > 0 ? ctx->leafBatchBvh2 : ctx->haveInstances ? 16 : ctx->leafBatch);
    const DeviceScene &s = ctx->scene;
    const int grid = int(ctx->poolGrid);
    const bool count = ctx->countTraversal;
    const bool flat = isFlat(ctx);
    const bool fused = flat && !ctx->haveForward && !ctx->haveMeshLight && ctx->fuseFlatOpt && !ctx->auxPass && !ctx->haveCylinder && ctx->tablesFit && !ctx->cameraFix;
    const bool runToCompletion = fused && !ctx->haveComplex && ctx->loopOpt;   // one launch renders the whole batch
    const size_t ldsBytes = traceLdsBytes(ctx, ctx->thrClosest);

    // optional per-launch timing: one event pair per kernel launch of a check interval, read back at the
    // interval's host sync (events live on ctx->stream, the stream the kernels are launched on)
    const bool timing = ctx->timeKernels;
    // (short batches: 8 -- every check drains all streams, ~0.2 ms in which nothing runs, and most of a short pass is its drain: 27 near-empty
    // iterations of ~150 us each for the 16-spp passes of the as-shipped materialtest, profiles/README.md; the price is <= 7 empty iterations
    // of ~55 us after the last path has ended)
    const int checkInterval = ctx->checkInterval > 0 ? ctx->checkInterval : (uint64_t(pp.total_items) >= 4ull*st.num_slots ? 16 : 8);
    const size_t evNeeded = size_t(checkInterval)*3*2*8;          // (every part's launches are timed: up to eight parts)
    if (timing) {
        while (ctx->evPool.size() < evNeeded) {
            hipEvent_t e = nullptr;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->evPool.push_back(e);
        }
    }
    size_t evUsed = 0;
    // (on the stream the launch goes to: ctx->launchStream is the part's own stream inside the split loop)
    auto ticMain = [&]() { if (timing) (void)hipEventRecord(ctx->evPool[evUsed++], ctx->launchStream); };
    auto tic = ticMain;

    // "streams" = N > 1: the workgroups (and with them the slots, queues and work items) are split into N parts that run the same
    // wavefront loop on N streams: kernels of different parts share the CUs (chooseThreads), and the drain tail of one part's kernel
    // overlaps the other parts' kernels
    // (materialtest 1280x720x256 / mesh1m 1920x1080x32 on one box: 2 parts 656 / 398 Msamples/s, 4 parts 666 / 412)
    // (instanced scenes: two parts in rounds 2-3; with the walk of the reference's instance tree four are 5 % faster -- profiles/r4_sweep_instances10k.jsonl)
    int parts = ctx->streamsOpt >= 2 ? ctx->streamsOpt : (ctx->streamsOpt == 1 || ctx->shortBatch) ? 1 : (!ctx->haveInstances || ctx->blocksPerCu >= 8) ? 4 : 1;
    if (fused || flat || st.records || grid < 2*parts || grid % parts != 0 || pp.total_items < uint32_t(2*parts)*PT_ITEM_GROUP)
        parts = 1;
    const bool split = parts > 1;
    PathState stPart[8] = {st, st, st, st, st, st, st, st};
    PassParams ppPart[8] = {pp, pp, pp, pp, pp, pp, pp, pp};
    hipStream_t streamOf[8] = {ctx->stream, ctx->partStream[0], ctx->partStream[1], ctx->partStream[2], ctx->partStream[3], ctx->partStream[4], ctx->partStream[5], ctx->partStream[6]};
    if (split) {
        const uint32_t groups = (pp.total_items + PT_ITEM_GROUP - 1)/PT_ITEM_GROUP;
        uint32_t itemBegin = 0;
        for (int k = 0; k < parts; ++k) {
            const uint32_t off = uint32_t(grid/parts)*uint32_t(k);
            for (uint32_t g = 0; g < PT_POOL_GROUPS; ++g)
                stPart[k].poolg[g] = st.poolg[g] + size_t(off)*st.slots_per_block*16u;
            stPart[k].bm = st.bm + size_t(off)*(st.slots_per_block >> 5);
            stPart[k].ctl = st.ctl + off;
            stPart[k].stats = st.stats + off;
            stPart[k].num_slots = st.num_slots/uint32_t(parts);
            const uint32_t itemEnd = k + 1 == parts ? pp.total_items : std::min(pp.total_items, uint32_t((uint64_t(groups)*(k + 1) + parts - 1)/parts)*PT_ITEM_GROUP);
            ppPart[k].item_begin = pp.item_begin + itemBegin;
            ppPart[k].total_items = itemEnd - itemBegin;
            itemBegin = itemEnd;
        }
    }
    HIP_TRY(ctx, hipMemsetAsync(st.partial, 0, size_t(pp.total_items)*sizeof(float4), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(st.live, 0, 4*sizeof(uint32_t), ctx->stream));
    if (split) {
        HIP_TRY(ctx, hipEventRecord(ctx->evMain, ctx->stream));            // (the memsets above come first for the other streams as well)
        for (int k = 0; k < parts; ++k) {
            if (k) HIP_TRY(ctx, hipStreamWaitEvent(streamOf[k], ctx->evMain, 0));
            hipLaunchKernelGGL(k_start, dim3(grid/parts), dim3(256), 0, streamOf[k], s, stPart[k], ppPart[k]);
        }
    } else {
        hipLaunchKernelGGL(k_start, dim3(grid), dim3(256), 0, ctx->stream, s, st, pp);
    }
    // k_tail runs the wide single-level kernels' bodies: scenes those kernels render, passes without visit counts (per-launch timing does
    // not see it: the few thousand rays it traces are in the counters, its one launch is in none of the three kernel classes)
    const bool tailEligible = ctx->tailOpt && !ctx->cameraFix && ctx->tablesFit && !flat && !ctx->haveInstances && wideClosest(ctx) && wideShadowRays(ctx) && ctx->decoupleOpt && !ctx->haveForward &&
                              (ctx->complexMask & TYPES_LATE) == 0 &&
                              !ctx->haveMeshLight && !ctx->haveMedia && !ctx->auxPass && !ctx->haveCylinder && !count && !st.records &&
                              st.slots_per_block <= PT_MAX_SLOTS_PER_BLOCK;
    const uint64_t tailThreshold = uint64_t(std::max<long long>(ctx->tailThreshold, 0));
    // (k_tail runs 256 threads whatever the depth of the tree: it is used only where a workgroup of it fits a CU -- its static LDS plus
    // the walk stacks of a very deep tree may not)
    bool tailFits = false;
    if (tailEligible) {
        int nb = 0;
        // (the conductor-family tail, tailCoat below, needs no more than the all-types one)
        const void *fn = pp.flags ? (ctx->haveSolids ? reinterpret_cast<const void *>(k_tail<(MASK_TAIL | FEAT_QMC), true>) : reinterpret_cast<const void *>(k_tail<(MASK_TAIL | FEAT_QMC), false>))
                                  : (ctx->haveSolids ? reinterpret_cast<const void *>(k_tail<MASK_TAIL, true>) : reinterpret_cast<const void *>(k_tail<MASK_TAIL, false>));
        tailFits = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, wideLdsBytes(ctx, 256)) == hipSuccess && nb >= 1;
    }
    // "fold_finish": k_finish rides in front of the next iteration's closest-hit launch (single-level scenes on the decoupled wide walk)
    const bool foldFinish = ctx->foldFinishOpt && !ctx->cameraFix && !flat && !ctx->haveInstances && wideClosest(ctx) && ctx->decoupleOpt;
    uint32_t iterTag = 1;                        // k_start publishes tag 1 when it queued anything
    bool first = true;
    int roundIters = checkInterval;              // launches of the wavefront loop between two host checks
        // the launches of one wavefront iteration over the workgroups [0, grid) of `st` (the whole pool, or one part of it)
        auto launchIteration = [&](const PathState &st, const PassParams &pp, int grid, uint32_t iterTag, bool timed, int part) {
            auto tic = [&]() { if (timed) ticMain(); };
            if (ctx->cameraFix)                  // (the fresh camera rays of the previous iteration's finish / of k_start, before they are traced)
                hipLaunchKernelGGL(k_camera_rays, dim3(grid), dim3(256), 0, ctx->launchStream, s, st, pp);
            tic();
            if (flat) {
                if (count) hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->launchStream, s, st);
                else       hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->launchStream, s, st);
            } else if (ctx->haveInstances && !wideClosest(ctx) && ctx->dynamicFetch && ctx->instDynOpt) {
                // the three-level walk with dynamic ray fetch
                const bool iw = instWide(ctx);               // the masters through the wide BVH (round 6)
                const size_t ldsDyn = iw ? instWideLdsBytes(ctx, ctx->thrClosest) : dynLdsBytes(ctx, ctx->thrClosest);
#define CLOSEST_DYN_INST(C, S) do { if (iw) hipLaunchKernelGGL((k_trace_closest_instw<C, S>), dim3(grid), dim3(ctx->thrClosest), ldsDyn, ctx->launchStream, s, st); \
                                    else hipLaunchKernelGGL((k_trace_closest_inst<C, S>), dim3(grid), dim3(ctx->thrClosest), ldsDyn, ctx->launchStream, s, st); } while (0)
                if (ctx->haveSolids) { if (count) CLOSEST_DYN_INST(true, true); else CLOSEST_DYN_INST(false, true); }
                else                 { if (count) CLOSEST_DYN_INST(true, false); else CLOSEST_DYN_INST(false, false); }
#undef CLOSEST_DYN_INST
            } else if (ctx->haveInstances && !wideClosest(ctx)) {
#define CLOSEST_INST(C, I) hipLaunchKernelGGL((k_trace_closest<C, false, I>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->launchStream, s, st)
                if (ctx->haveSolids) { if (count) CLOSEST_INST(true, 1); else CLOSEST_INST(false, 1); }
                else                 { if (count) CLOSEST_INST(true, 2); else CLOSEST_INST(false, 2); }
#undef CLOSEST_INST
            } else if (wideClosest(ctx)) {
                const size_t ldsWide = wideLdsBytes(ctx, ctx->thrClosest);
#define CLOSEST_WIDE(C, S, I, D) hipLaunchKernelGGL((k_trace_closest_wide<C, S, I, D>), dim3(grid), dim3(ctx->thrClosest), ldsWide, ctx->launchStream, s, st)
                if (ctx->haveInstances) {
                    if (ctx->haveSolids) { if (count) CLOSEST_WIDE(true, true, true, false); else CLOSEST_WIDE(false, true, true, false); }
                    else                 { if (count) CLOSEST_WIDE(true, false, true, false); else CLOSEST_WIDE(false, false, true, false); }
                } else if (ctx->decoupleOpt && foldFinish) {
                    // (k_finish's work of the previous iteration in front of the walk, pt_wavefront.h)
#define CLOSEST_FIN(C, S, E) hipLaunchKernelGGL((k_finish_trace_closest_wide<C, S, E>), dim3(grid), dim3(ctx->thrClosest), ldsWide, ctx->launchStream, s, st, pp)
                    // ("finish_lean" = 1, an experiment: passes without flags -- uniform sampler, no records, no auxiliary or per-sample output, pinhole, no media: the metric's --
                    // finish through nextPath's lean variant, 4 220 instead of 6 436 instructions in the kernel; 0.45 % SLOWER on the metric, three alternations: the walk's
                    // code around it comes out differently.  Off by default; profiles/r6_ab_finish_lean.txt)
                    const bool leanFinish = ctx->finishLeanOpt && pp.flags == 0u && pp.rec_count == nullptr && !count;
                    if (leanFinish)           { if (ctx->haveSolids) CLOSEST_FIN(false, true, false); else CLOSEST_FIN(false, false, false); }
                    else if (ctx->haveSolids) { if (count) CLOSEST_FIN(true, true, true); else CLOSEST_FIN(false, true, true); }
                    else                      { if (count) CLOSEST_FIN(true, false, true); else CLOSEST_FIN(false, false, true); }
#undef CLOSEST_FIN
                } else if (ctx->decoupleOpt) {
                    if (ctx->haveSolids) { if (count) CLOSEST_WIDE(true, true, false, true); else CLOSEST_WIDE(false, true, false, true); }
                    else                 { if (count) CLOSEST_WIDE(true, false, false, true); else CLOSEST_WIDE(false, false, false, true); }
                } else {
                    if (ctx->haveSolids) { if (count) CLOSEST_WIDE(true, true, false, false); else CLOSEST_WIDE(false, true, false, false); }
                    else                 { if (count) CLOSEST_WIDE(true, false, false, false); else CLOSEST_WIDE(false, false, false, false); }
                }
#undef CLOSEST_WIDE
            } else {
                if (ctx->dynamicFetch) {
                    const size_t ldsDyn = dynLdsBytes(ctx, ctx->thrClosest);
#define CLOSEST_DYN(C, S) hipLaunchKernelGGL((k_trace_closest_dyn<C, S>), dim3(grid), dim3(ctx->thrClosest), ldsDyn, ctx->launchStream, s, st)
                    if (ctx->haveSolids) { if (count) CLOSEST_DYN(true, true); else CLOSEST_DYN(false, true); }
                    else                 { if (count) CLOSEST_DYN(true, false); else CLOSEST_DYN(false, false); }
#undef CLOSEST_DYN
                } else {
                    if (count) hipLaunchKernelGGL((k_trace_closest<true, false>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->launchStream, s, st);
                    else       hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(grid), dim3(ctx->thrClosest), ldsBytes, ctx->launchStream, s, st);
                }
            }
            tic(); tic();
            // The shading classes of an iteration (the escaped paths, Q_MISS; 0; the classes 1 .. 3 that occur) consume queues of their own and OR what they append
            // into the workgroup's bitmaps (k_shade: CONCURRENT), so their launches MAY run side by side on three streams ("class_streams"
            // option).  Measured slower than one after the other (materialtest 735-800 against 825-830 Msamples/s, mesh1m 409 against 508,
            // for 2 to 24 hardware queues): with four parts in flight the chip is not short of independent launches.
            auto shadeClass = [&](int cls) {
                const bool simple = cls == 0 || cls == CLS_MISS || cls == CLS_0_AND_MISS;
                if (ctx->mediaSimple && ctx->mediaLeanOpt && !ctx->auxPass && !ctx->haveCylinder && !ctx->haveMeshLight)
                    launchShadeVariant<MASK_MEDIA, 0>(ctx, grid, st, pp, cls);                      // media scenes with simple surfaces: no scratch
                else if (ctx->haveMedia || ctx->auxPass || ctx->haveCylinder) launchShadeVariant<BSDF_MASK_ALL, 0>(ctx, grid, st, pp, cls);   // the one variant with FEAT_MEDIA / FEAT_AUX / FEAT_CYLINDER
                else if (ctx->haveMeshLight) launchShade<MASK_FULL>(ctx, grid, st, pp, cls);   // the only variant with mesh-emitter sampling (every BSDF type)
                else if (ctx->haveInstances && simple && ctx->instSimpleOpt) launchShade<MASK_SIMPLE_INST>(ctx, grid, st, pp, cls);   // Lambert / escaped paths of instanced scenes
                else if (ctx->haveInstances && (simple || !ctx->instSimpleOpt)) launchShade<MASK_FULL>(ctx, grid, st, pp, cls);
                else if (simple) {               // the escaped paths run the class-0 variant: its surface code never runs there, so the launch is short
                    if (ctx->leanScene) launchShade<MASK_LEAN>(ctx, grid, st, pp, cls);
                    else                launchShade<MASK_SIMPLE>(ctx, grid, st, pp, cls);
                }
                else launchComplexClass<0>(ctx, grid, st, pp, cls);
            };
            hipStream_t mainStream = ctx->launchStream;
            if (ctx->classStreamsOpt != 0) {
                hipStream_t *aux = ctx->classStream[part];
                (void)hipEventRecord(ctx->evFork[part], mainStream);
                const int nAux = ctx->haveComplex ? 2 : 1;
                for (int a = 0; a < nAux; ++a) {
                    (void)hipStreamWaitEvent(aux[a], ctx->evFork[part], 0);
                    ctx->launchStream = aux[a];
                    shadeClass(a == 0 ? CLS_MISS : 0);
                    (void)hipEventRecord(ctx->evJoin[part][a], aux[a]);
                }
                ctx->launchStream = mainStream;
                if (!ctx->haveComplex) shadeClass(0);             // the longest launches stay on the part's own stream
                for (int c = 1; c < PT_NUM_CLASSES; ++c)
                    if (ctx->classPresent[c]) shadeClass(c);
                for (int a = 0; a < nAux; ++a)
                    (void)hipStreamWaitEvent(mainStream, ctx->evJoin[part][a], 0);
            } else {
                // (class 0 and the escaped paths run the same variant: one launch over both queues, "merge_miss" = 0 keeps them apart)
                if (ctx->mergeMissOpt) shadeClass(CLS_0_AND_MISS);
                else { shadeClass(CLS_MISS); shadeClass(0); }
                for (int c = 1; c < PT_NUM_CLASSES; ++c)
                    if (ctx->classPresent[c]) shadeClass(c);
            }
            tic(); tic();
            const bool finish = count ? launchShadow<true>(ctx, grid, st, pp, iterTag) : launchShadow<false>(ctx, grid, st, pp, iterTag);
            tic();
            (void)finish;                        // (always: the shading launches leave their finished paths to k_finish as well)
            if (!foldFinish)                     // (folded: the next iteration's closest-hit launch, or finishParts before a host check)
                hipLaunchKernelGGL(k_finish, dim3(grid), dim3(256), 0, ctx->launchStream, s, st, pp, iterTag);
        };
        // folded k_finish: the last iteration's finish as a launch of its own -- before the host reads the liveness word, before k_tail
        int partStream[8] = {0, 1, 2, 3, 4, 5, 6, 7};   // the stream a part's last iteration ran on ("rotate_streams")
        bool partRan[8] = {false, false, false, false, false, false, false, false};
        auto finishParts = [&](uint32_t tag) {
            for (int k = 0; k < parts; ++k)
                hipLaunchKernelGGL(k_finish, dim3(grid/parts), dim3(256), 0, split ? streamOf[partStream[k]] : ctx->stream, s, split ? stPart[k] : st, split ? ppPart[k] : pp, tag);
        };
    for (;;) {
        evUsed = 0;
        {
            // an abort request that arrived since the last check (or before the pass's first launch): make sure the device
            // word is set -- in stream order, so the launches below see it and drain their slots
            for (int k = 1; k < parts; ++k) {   // the main stream waits for the other parts before it reads the flag
                HIP_TRY(ctx, hipEventRecord(ctx->evPart[k - 1], streamOf[k]));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->evPart[k - 1], 0));
            }
            if (ctx->abortRequested.load(std::memory_order_acquire))
                HIP_TRY(ctx, hipMemsetAsync(ctx->abortFlagDev, 0xFF, sizeof(uint32_t), ctx->stream));
            if (tailEligible) hipLaunchKernelGGL(k_live_slots, dim3(1), dim3(256), 0, ctx->stream, st, uint32_t(grid));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->hostLive, st.live, 2*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (timing && !first) {
                double *acc[3] = {&ctx->counters.ms_trace_closest, &ctx->counters.ms_shade, &ctx->counters.ms_trace_shadow};
                // (split loop: EVERY part's launches are bracketed, each on its own stream -- three pairs per part and iteration, in launch
                // order.  Rounds 2-4 timed part 0 only and counted the others "as long on average": they are not -- the parts whose
                // streams the hardware queues serve later run ~45 % longer launches (rocprofv3's kernel trace, profiles/README.md) --, so the
                // per-kernel averages were part 0's, a fifth below the mean)
                size_t pairs = size_t(roundIters)*3*size_t(parts);
                for (size_t k = 0; k < pairs; ++k) {
                    float ms = 0.0f;
                    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->evPool[2*k], ctx->evPool[2*k + 1]));
                    *acc[k % 3] += double(ms);
                }
                const int perIter = parts;
                ctx->counters.launches_trace_closest += roundIters*perIter;
                ctx->counters.launches_trace_shadow += roundIters*perIter;
                ctx->counters.launches_shade += roundIters*perIter;
            }
            if (ctx->hostLive[0] != iterTag)
                break;                           // the last iteration left every extension queue empty
            if (tailFits && uint64_t(ctx->hostLive[1]) <= tailThreshold) {
                // few paths left: each part of the pool finishes in one launch of k_tail (its workgroups iterate on their own)
                const size_t ldsTail = wideLdsBytes(ctx, 256);
                uint32_t classes = 1u;
                for (int c = 1; c < PT_NUM_CLASSES; ++c)
                    if (ctx->classPresent[c]) classes |= 1u << c;
                for (int k = 0; k < parts; ++k) {
                    const PathState &sp = split ? stPart[k] : st;
                    const PassParams &ppk = split ? ppPart[k] : pp;
                    const hipStream_t stream = split ? streamOf[k] : ctx->stream;
                    if (k) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->evMain, 0));   // (after the main stream's check above)
                    else if (split) HIP_TRY(ctx, hipEventRecord(ctx->evMain, ctx->stream));
#define TAIL_LAUNCH(M, S) hipLaunchKernelGGL((k_tail<M, S>), dim3(grid/parts), dim3(256), ldsTail, stream, s, sp, ppk, classes)
                    // scenes of Lambert / null and conductor-family materials only (the metric's): the narrower instantiation, "tail_family" = 0 for the A/B
                    const bool tailCoat = ctx->tailFamilyOpt && !ctx->haveSolids && !ctx->classPresent[2] && !ctx->classPresent[3];
                    if (tailCoat) { if (pp.flags) TAIL_LAUNCH((MASK_COAT | FEAT_QMC), false); else TAIL_LAUNCH(MASK_COAT, false); }
                    else if (pp.flags) { if (ctx->haveSolids) TAIL_LAUNCH((MASK_TAIL | FEAT_QMC), true); else TAIL_LAUNCH((MASK_TAIL | FEAT_QMC), false); }
                    else          { if (ctx->haveSolids) TAIL_LAUNCH(MASK_TAIL, true); else TAIL_LAUNCH(MASK_TAIL, false); }
#undef TAIL_LAUNCH
                }
                for (int k = 1; k < parts; ++k) {
                    HIP_TRY(ctx, hipEventRecord(ctx->evPart[k - 1], streamOf[k]));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->evPart[k - 1], 0));
                }
                ctx->counters.tail_launches++;
                if (std::getenv("TGHIP_VERBOSE")) {
                    const auto t0 = std::chrono::steady_clock::now();
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    std::fprintf(stderr, "[tghip]   tail kernel after %llu iterations with %u paths alive: %.2f ms\n", (unsigned long long)(iterTag - 1), ctx->hostLive[1],
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                }
                break;
            }
        }
        first = false;
        roundIters = runToCompletion ? 1 : checkInterval;
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
                for (int c = 1; c < PT_NUM_CLASSES; ++c)
                    if (ctx->classPresent[c]) launchComplexClass<FUSE_SHADOW>(ctx, grid, st, ppi, c);
                tic(); tic(); tic();
                ctx->counters.iterations++;
                continue;
            }
            if (split) {
                // "rotate_streams": part k's iteration i runs on stream (k + i) mod parts.  The hardware queues do not share the chip evenly --
                // the parts on the queues served later run ~45 % longer launches (profiles/README.md: "by hardware queue") and finish their
                // share of the work items last --; rotated, every part spends the same time on every queue.
                for (int k = 0; k < parts; ++k) {
                    const int sIdx = ctx->rotateStreamsOpt ? int((uint32_t(k) + iterTag) % uint32_t(parts)) : k;
                    ctx->launchStream = streamOf[sIdx];
                    if (ctx->rotateStreamsOpt) {
                        if (partRan[k]) (void)hipStreamWaitEvent(ctx->launchStream, ctx->evRot[k], 0);    // the part's previous iteration, on another stream
                        partStream[k] = sIdx;
                    }
                    launchIteration(stPart[k], ppPart[k], grid/parts, iterTag, true, k);
                    if (ctx->rotateStreamsOpt) { (void)hipEventRecord(ctx->evRot[k], ctx->launchStream); partRan[k] = true; }
                }
                ctx->launchStream = ctx->stream;
            } else {
                launchIteration(st, pp, grid, iterTag, true, 0);
            }
            ctx->counters.iterations++;
        }
        if (foldFinish && !fused)
            finishParts(iterTag);
    }
    float *sum = ctx->extSum ? ctx->extSum : ctx->fbSum;
    uint32_t *cnt = ctx->extCount ? ctx->extCount : ctx->fbCount;
    if (pp.rec_sorted) hipLaunchKernelGGL(k_resolve_records, dim3((pp.num_sorted*16u + 255)/256), dim3(256), 0, ctx->stream, st, pp, sum, cnt);
    else               hipLaunchKernelGGL(k_resolve, dim3((pp.pix_slots + 255)/256), dim3(256), 0, ctx->stream, st, pp, sum, cnt);
    HIP_TRY(ctx, hipGetLastError());
    return ctx->abortRequested.load(std::memory_order_acquire) ? TGHIP_E_ABORTED : TGHIP_OK;
}

// PassParams::rec_hint of a record pass: hint[g] = the chunk item 64 g lies in -- the last c < chunks with start[c] <= 64 g (the chunk starts never
// decrease; nextPath walks on from there while its item is >= start[c + 1], which ends at the sentinel behind the last chunk at the latest)
__global__ void k_record_hints(const uint32_t *start, uint32_t chunks, uint32_t *hint, uint32_t n)
{
    const uint32_t g = blockIdx.x*blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint64_t item = uint64_t(g)*64u;
    uint32_t lo = 0, hi = chunks;                 // invariant: start[lo] <= item, and start[hi] > item or hi == chunks
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (uint64_t(start[mid]) <= item) lo = mid; else hi = mid;
    }
    hint[g] = lo;
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
    if (pass->flags & ~(TGHIP_PASS_SOBOL | TGHIP_PASS_RECORDS | TGHIP_PASS_AUX | TGHIP_PASS_SAMPLES)) { ctx->error = "unknown pass flags"; return TGHIP_E_INVALID; }
    if ((pass->flags & TGHIP_PASS_SOBOL) && (!ctx->scene.sobol || !pass->tile_seeds)) {
        ctx->error = "TGHIP_PASS_SOBOL needs sobol_matrices in the scene description and tile_seeds in the pass";
        return TGHIP_E_INVALID;
    }
    if ((pass->record_count != nullptr) != (pass->record_index != nullptr) || (pass->record_count && !(pass->flags & TGHIP_PASS_RECORDS))) {
        ctx->error = "record_index and record_count go together and need TGHIP_PASS_RECORDS";
        return TGHIP_E_INVALID;
    }
    if ((pass->flags & TGHIP_PASS_SAMPLES) && (pass->record_count || uint64_t(ctx->width)*ctx->height*(pass->spp_end - pass->spp_begin)*3u >= (1ull << 31))) {
        ctx->error = "TGHIP_PASS_SAMPLES keeps W*H*spp*3 floats (< 2^31) and does not combine with per-record sample counts";
        return TGHIP_E_INVALID;
    }
    ctx->abortRequested.store(false, std::memory_order_release);   // the only place a request is cleared
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
    const bool verbose = std::getenv("TGHIP_VERBOSE") != nullptr;
    const auto tWait0 = std::chrono::steady_clock::now();
    auto msSince = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const TgHipPassDesc pass = ctx->pendingPass;
    const uint32_t shardCount = pass.shard_count ? pass.shard_count : 1;
    const uint32_t w = ctx->width, h = ctx->height;
    const uint32_t tilesX = (w + 15)/16, tilesY = (h + 15)/16;
    const uint32_t numTiles = tilesX*tilesY;
    // the shard's tiles (tghip_tile_owner: a diagonal interleave), row-major; the device indexes this list (PassParams::owned_tiles)
    const uint32_t shardSkew = tghip_shard_skew(shardCount);
    auto ownsTile = [&](uint32_t tx, uint32_t ty) { return tghip_tile_owner(tx, ty, shardCount) == pass.shard_index; };
    uint32_t ownedTiles = numTiles;
    if (shardCount > 1) {
        if (ctx->ownedKey[0] != w || ctx->ownedKey[1] != h || ctx->ownedKey[2] != pass.shard_index || ctx->ownedKey[3] != shardCount) {
            std::vector<uint32_t> &list = ctx->hostOwnedTiles;
            list.clear();
            for (uint32_t ty = 0; ty < tilesY; ++ty)
                for (uint32_t tx = 0; tx < tilesX; ++tx)
                    if (ownsTile(tx, ty)) list.push_back(tx + ty*tilesX);
            if (ctx->ownedCap < list.size()) {
                if (ctx->dOwnedTiles) (void)hipFree(ctx->dOwnedTiles);
                ctx->dOwnedTiles = nullptr; ctx->ownedCap = 0;
                HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dOwnedTiles), std::max<size_t>(list.size(), 1)*sizeof(uint32_t)));
                ctx->ownedCap = list.size();
            }
            if (!list.empty())
                HIP_TRY(ctx, hipMemcpyAsync(ctx->dOwnedTiles, list.data(), list.size()*sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // (the list is host memory of the context; keep it simple: rare)
            ctx->ownedKey[0] = w; ctx->ownedKey[1] = h; ctx->ownedKey[2] = pass.shard_index; ctx->ownedKey[3] = shardCount;
        }
        ownedTiles = uint32_t(ctx->hostOwnedTiles.size());
    }
    uint32_t spp = pass.spp_end - pass.spp_begin, sppBegin = pass.spp_begin;
    PassParams base{};
    base.shard_skew = shardSkew;
    base.owned_tiles = shardCount > 1 ? ctx->dOwnedTiles : nullptr;
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
            bool owned = ownsTile(rx >> 2, ry >> 2);
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
    // Samples per work item: 4 when the pass is long (256 spp at 720p: 59 M items over 8 M slots) -- but a slot works its item's samples
    // off one after the other, so a SHORT pass cut into 4-sample items is a few long sequential chains on a half-empty pool: the 16-spp
    // passes of the as-shipped materialtest (3.7 M items) ran at half the throughput of one 256-spp pass.  Such passes -- and one rank's
    // share of a multi-GPU render -- get smaller items, down to one sample each, so that the items outnumber the slots about twice.
    uint32_t chunk = ctx->auxPass ? std::max(spp, 1u) : uint32_t(std::max(ctx->chunkSamples, 0));
    if (chunk == 0) {
        uint64_t passSamples = 0;
        if (pass.flags & TGHIP_PASS_RECORDS) {
            uint64_t c16 = 0;
            for (uint32_t r = 0; r < numRecords; ++r) {
                const uint32_t rx = r % base.variance_w, ry = r/base.variance_w;
                if (ownsTile(rx >> 2, ry >> 2)) c16 += ctx->hostRecCount[r];
            }
            passSamples = c16*16u;
        } else {
            passSamples = uint64_t(ownedTiles)*256u*spp;
        }
        // One sample or four, nothing in between: k_resolve adds one-sample items up in groups of four, so both choices -- and with them
        // the unsharded pass and every shard of it, which pick by their OWN sample counts -- round a pixel's sum alike (pt_wavefront.h:
        // k_resolve; tests/test_gpu_parity.py::test_shards_and_batches_round_like_the_whole_pass).  An explicit "chunk_samples" of 2 or 3
        // gives up that property, not the result's validity.
        chunk = (passSamples >> 24) >= 3 ? 4u : 1u;
    }
    const uint32_t group = chunk == 1u ? 4u : 1u;
    if (ctx->auxPass && !ctx->dAux) {
        const size_t npix = size_t(w)*h;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dAux), npix*sizeof(TgHipAuxPixel)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dAux, 0, npix*sizeof(TgHipAuxPixel), ctx->stream));
    }
    base.aux = ctx->dAux;
    if (pass.flags & TGHIP_PASS_SAMPLES) {
        const size_t need = size_t(w)*h*spp*3u;
        if (ctx->samplesCap < need) {
            if (ctx->dSamples) (void)hipFree(ctx->dSamples);
            ctx->dSamples = nullptr; ctx->samplesCap = 0;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dSamples), need*sizeof(float)));
            ctx->samplesCap = need;
        }
        HIP_TRY(ctx, hipMemsetAsync(ctx->dSamples, 0, need*sizeof(float), ctx->stream));
        ctx->samplesFloats = need;
        base.samples = ctx->dSamples; base.samples_begin = sppBegin; base.samples_spp = spp;
    }
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
            if (ownsTile(rx >> 2, ry >> 2) && cnt[r] > 0)
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
        // (the hint table -- for every 64 items the chunk their first one lies in, 230 k entries for a 16-spp pass at 720p -- is filled by a launch
        // from the chunk starts: building and uploading it here was a third of the host's set-up time between two passes)
        const size_t hintEntries = size_t((recordItems + 63)/64) + 1;
        (void)hint;
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
        if (ctx->hintCap < hintEntries) {
            if (ctx->dHint) (void)hipFree(ctx->dHint);
            ctx->dHint = nullptr; ctx->hintCap = 0;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dHint), hintEntries*sizeof(uint32_t)));
            ctx->hintCap = hintEntries;
        }
        hipLaunchKernelGGL(k_record_hints, dim3(uint32_t((hintEntries + 255)/256)), dim3(256), 0, ctx->stream, ctx->dChunkStart, numChunks, ctx->dHint, uint32_t(hintEntries));
        HIP_TRY(ctx, hipGetLastError());
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
        chunksPerBatch = std::max(group, chunksPerBatch/group*group);          // whole groups of k_resolve
        if (uint64_t(ownedTiles)*256*chunksPerBatch > maxItems)
            tilesPerBatch = uint32_t(std::max<uint64_t>(1, maxItems/(256ull*chunksPerBatch)));
    }
    // record passes: consecutive ranges of the gap-free enumeration that hold whole groups of chunks (the items of chunk c are
    // [start[c], start[c + 1])), as many groups as fit maxItems, at least one
    std::vector<uint64_t> recordBatchEnd;
    if (recordPass) {
        const std::vector<uint32_t> &start = ctx->hostChunkStart;
        uint64_t begin = 0;
        for (uint32_t c = 0; c < chunksAll; c += group) {
            const uint64_t end = start[std::min(c + group, chunksAll)];
            if (end - begin > maxItems && start[c] > begin) { recordBatchEnd.push_back(start[c]); begin = start[c]; }
        }
        recordBatchEnd.push_back(recordItems);
    }
    uint64_t recordBatchMax = 0;
    for (size_t i = 0; i < recordBatchEnd.size(); ++i)
        recordBatchMax = std::max(recordBatchMax, recordBatchEnd[i] - (i ? recordBatchEnd[i - 1] : 0));
    const uint64_t batchItems = recordPass ? recordBatchMax : uint64_t(tilesPerBatch)*256*chunksPerBatch;
    {
        // A pass whose work items fill less than half the pool never refills a slot: it is one long drain, every iteration of which
        // pays the fixed cost of its launches.  Such passes (the 16-spp passes of the as-shipped materialtest: 3.7 M items) run on ONE
        // stream with 4 workgroups per CU -- 431 against 372 Msamples/s with four parts.  (chooseThreads picks grid and workgroup sizes)
        const bool shortBatch = 2u*batchItems <= uint64_t(ctx->maxSlots);   // (one rank's share of an 8-GPU render of the metric's workload, 7.4 M items, stays on four parts)
        if (shortBatch != ctx->shortBatch) {
            ctx->shortBatch = shortBatch;
            chooseThreads(ctx);
        }
    }
    uint64_t wantSlots = std::min<uint64_t>(uint64_t(ctx->maxSlots), batchItems);
    {
        // the run-to-completion kernel (runBatch) gives every thread its own slots for the whole launch: one slot per
        // thread keeps the whole path state (112 B x 0.5 M slots) inside the Infinity Cache -- measured +5 % over four
        const bool flat = isFlat(ctx);
        const bool loop = flat && !ctx->haveForward && !ctx->haveMeshLight && ctx->fuseFlatOpt && !ctx->cameraFix && !ctx->haveComplex && ctx->loopOpt && !ctx->auxPass && !ctx->haveCylinder && ctx->tablesFit;
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
    // the device word still holds the previous pass's abort, if any; a request for THIS pass is re-mirrored by runBatch
    HIP_TRY(ctx, hipMemsetAsync(ctx->abortFlagDev, 0, sizeof(uint32_t), ctx->stream));

    const double msSetup = msSince(tWait0);
    const unsigned long long itersBefore = ctx->counters.iterations;
    const auto tLoop0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipEventRecord(ctx->evA, ctx->stream));
    for (size_t bi = 0; recordPass && bi < recordBatchEnd.size() && rc == TGHIP_OK; ++bi) {
        const uint64_t w0 = bi ? recordBatchEnd[bi - 1] : 0;
        PassParams pp = base;
        pp.spp_begin = 0; pp.spp_end = spp; pp.seed = pass.seed;
        pp.chunk = chunk;
        pp.group = group;
        pp.chunks = chunksAll;
        pp.pix_slots = 1;                            // unused by the record enumeration
        pp.item_base = uint32_t(w0);
        pp.total_items = uint32_t(recordBatchEnd[bi] - w0);
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
            pp.group = group;
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
    if (verbose)
        std::fprintf(stderr, "[tghip] pass spp [%u, %u) flags %u: %llu items of %u samples in batches of %llu, %u slots, short %d; setup %.2f ms, loop %.2f ms (%llu iterations, device %.2f ms)\n",
                     pass.spp_begin, pass.spp_end, pass.flags, (unsigned long long)(recordPass ? recordItems : uint64_t(ownedTiles)*256*chunksAll), chunk,
                     (unsigned long long)batchItems, ctx->pool.num_slots, int(ctx->shortBatch), msSetup, msSince(tLoop0), (unsigned long long)(ctx->counters.iterations - itersBefore), double(ms));
    return ctx->passResult = rc;
}

int tghip_abort(tghip_ctx *ctx)
{
    if (!ctx) return TGHIP_E_INVALID;
    // The request itself is host state: tghip_wait looks at it before every batch and at every liveness check, so it holds
    // even when no kernel of the pass has been launched yet.  The device word (polled whenever a slot asks for new work) is
    // written from a stream of its own so that the copy does not queue behind the running pass.
    ctx->abortRequested.store(true, std::memory_order_release);
    std::lock_guard<std::mutex> lock(ctx->abortMutex);
    static const uint32_t one = 1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMemcpyAsync(ctx->abortFlagDev, &one, sizeof(one), hipMemcpyHostToDevice, ctx->abortStream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->abortStream);
    if (e != hipSuccess) return TGHIP_E_HIP;    // (ctx->error belongs to the thread driving the pass)
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

int tghip_download_samples(tghip_ctx *ctx, float *rgb, size_t nfloats)
{
    if (!ctx || !ctx->haveScene) return TGHIP_E_NOSCENE;
    int rc = tghip_wait(ctx);
    if (rc != TGHIP_OK && rc != TGHIP_E_ABORTED) return rc;
    if (!rgb || !ctx->dSamples || nfloats != ctx->samplesFloats) { ctx->error = "no TGHIP_PASS_SAMPLES pass of that size has been rendered"; return TGHIP_E_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(rgb, ctx->dSamples, nfloats*sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGHIP_OK;
}

extern "C++" {
namespace {
// RCCL through dlopen: single-GPU users of this library never load it
struct RcclApi {
    std::mutex mutex;
    bool tried = false;
    void *lib = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;   // one communicator set per device list, kept for the life of the process
    bool load()
    {
        if (tried) return lib != nullptr;
        tried = true;
        // An RCCL the process already has comes first (RTLD_NOLOAD matches by SONAME): a host program that brought its own -- PyTorch-ROCm ships
        // one next to libtorch -- and this library must not end up with two copies whose exported symbols resolve into each other.
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return false;
        commInitAll = reinterpret_cast<decltype(commInitAll)>(dlsym(lib, "ncclCommInitAll"));
        reduce = reinterpret_cast<decltype(reduce)>(dlsym(lib, "ncclReduce"));
        groupStart = reinterpret_cast<decltype(groupStart)>(dlsym(lib, "ncclGroupStart"));
        groupEnd = reinterpret_cast<decltype(groupEnd)>(dlsym(lib, "ncclGroupEnd"));
        errorString = reinterpret_cast<decltype(errorString)>(dlsym(lib, "ncclGetErrorString"));
        getUniqueId = reinterpret_cast<decltype(getUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        commInitRank = reinterpret_cast<decltype(commInitRank)>(dlsym(lib, "ncclCommInitRank"));
        commDestroy = reinterpret_cast<decltype(commDestroy)>(dlsym(lib, "ncclCommDestroy"));
        if (!commInitAll || !reduce || !groupStart || !groupEnd || !errorString || !getUniqueId || !commInitRank || !commDestroy) { dlclose(lib); lib = nullptr; }
        return lib != nullptr;
    }
} g_rccl;
} // namespace
} // extern "C++"

int tghip_reduce_framebuffers(tghip_ctx *const *ctxs, int n, int root, float *rgb_sum, uint32_t *count, size_t npixels)
{
    if (!ctxs || n < 1 || root < 0 || root >= n) return TGHIP_E_INVALID;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return TGHIP_E_INVALID;
    tghip_ctx *rc = ctxs[root];
    std::vector<int> devices;
    for (int i = 0; i < n; ++i) {
        tghip_ctx *c = ctxs[i];
        if (c->failReduce) { rc->error = "tghip_reduce_framebuffers: forced failure (the \"fail_reduce\" option)"; return TGHIP_E_HIP; }
        if (!c->haveScene) { rc->error = "tghip_reduce_framebuffers: a context has no scene"; return TGHIP_E_NOSCENE; }
        if (c->width != rc->width || c->height != rc->height) { rc->error = "tghip_reduce_framebuffers: the contexts render different images"; return TGHIP_E_INVALID; }
        for (int d : devices)
            if (d == c->device) { rc->error = "tghip_reduce_framebuffers: two contexts on one device (RCCL wants one rank per device; sum on the host instead)"; return TGHIP_E_UNSUPPORTED; }
        devices.push_back(c->device);
        int w = tghip_wait(c);
        if (w != TGHIP_OK && w != TGHIP_E_ABORTED) return w;
    }
    if (npixels != size_t(rc->width)*rc->height) { rc->error = "pixel count mismatch"; return TGHIP_E_INVALID; }
    HIP_TRY(rc, hipSetDevice(rc->device));
    if (rc->redCap < npixels) {                  // (first call, or the context was re-uploaded with a larger image since)
        if (rc->redSum) (void)hipFree(rc->redSum);
        if (rc->redCount) (void)hipFree(rc->redCount);
        rc->redSum = nullptr; rc->redCount = nullptr; rc->redCap = 0;
        HIP_TRY(rc, hipMalloc(reinterpret_cast<void **>(&rc->redSum), npixels*3*sizeof(float)));
        HIP_TRY(rc, hipMalloc(reinterpret_cast<void **>(&rc->redCount), npixels*sizeof(uint32_t)));
        rc->redCap = npixels;
    }

    std::lock_guard<std::mutex> lock(g_rccl.mutex);
    if (!g_rccl.load()) { rc->error = "tghip_reduce_framebuffers: librccl.so could not be loaded"; return TGHIP_E_UNSUPPORTED; }
    auto ncclTry = [&](ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return true;
        rc->error = std::string(what) + ": " + g_rccl.errorString(r);
        return false;
    };
    std::vector<ncclComm_t> &comms = g_rccl.comms[devices];
    if (comms.empty()) {
        comms.resize(size_t(n));
        if (!ncclTry(g_rccl.commInitAll(comms.data(), n, devices.data()), "ncclCommInitAll")) { g_rccl.comms.erase(devices); return TGHIP_E_HIP; }
    }
    // two reductions per rank (radiance sums, sample counts) in one group; every rank's calls run on its own stream
    if (!ncclTry(g_rccl.groupStart(), "ncclGroupStart")) { g_rccl.comms.erase(devices); return TGHIP_E_HIP; }
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
        tghip_ctx *c = ctxs[i];
        const float *sum = c->extSum ? c->extSum : c->fbSum;
        const uint32_t *cnt = c->extCount ? c->extCount : c->fbCount;
        ok = ncclTry(g_rccl.reduce(sum, i == root ? rc->redSum : nullptr, npixels*3, ncclFloat32, ncclSum, root, comms[size_t(i)], c->stream), "ncclReduce")
          && ncclTry(g_rccl.reduce(cnt, i == root ? rc->redCount : nullptr, npixels, ncclUint32, ncclSum, root, comms[size_t(i)], c->stream), "ncclReduce");
    }
    if (!ncclTry(g_rccl.groupEnd(), "ncclGroupEnd") || !ok) { g_rccl.comms.erase(devices); return TGHIP_E_HIP; }   // (a communicator set that failed is not reused)
    for (int i = 0; i < n; ++i) {
        HIP_TRY(rc, hipSetDevice(ctxs[i]->device));
        HIP_TRY(rc, hipStreamSynchronize(ctxs[i]->stream));
    }
    HIP_TRY(rc, hipSetDevice(rc->device));
    if (rgb_sum) HIP_TRY(rc, hipMemcpyAsync(rgb_sum, rc->redSum, npixels*3*sizeof(float), hipMemcpyDeviceToHost, rc->stream));
    if (count) HIP_TRY(rc, hipMemcpyAsync(count, rc->redCount, npixels*sizeof(uint32_t), hipMemcpyDeviceToHost, rc->stream));
    HIP_TRY(rc, hipStreamSynchronize(rc->stream));
    return TGHIP_OK;
}

// the libm restatements exactly as the shading kernels call them (pt_math.h)
__global__ void k_debug_libm(int fn, const float *x, float *y, uint32_t n)
{
    const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool two = fn == TGHIP_LIBM_ATAN2F || fn == TGHIP_LIBM_POWF;
    const float v = two ? x[2u*i] : x[i], v2 = two ? x[2u*i + 1u] : 0.0f;
    float s, c, r;
    switch (fn) {
    case TGHIP_LIBM_ATAN2F: r = atan2fH(v, v2); break;
    case TGHIP_LIBM_POWF: r = powfH(v, v2); break;
    case TGHIP_LIBM_CBRTF: r = cbrtfH(v); break;
    case TGHIP_LIBM_EMBREE_RCP: r = embreeRcp(v); break;
    case TGHIP_LIBM_RCPPS: r = rcppsIntel(v); break;
    case TGHIP_LIBM_TANF: r = tanfH(v); break;
    case TGHIP_LIBM_SINF: r = sinfH(v); break;
    case TGHIP_LIBM_COSF: r = cosfH(v); break;
    case TGHIP_LIBM_LOGF: r = logfH(v); break;
    case TGHIP_LIBM_EXPF: r = expfH(v); break;
    case TGHIP_LIBM_SINCOS_SIN: sincosfH(v, s, c); r = s; break;
    case TGHIP_LIBM_SINCOS_COS: sincosfH(v, s, c); r = c; break;
    default: r = acosfExact(v); break;
    }
    y[i] = r;
}

// ... and the double-precision ones of the atmospheric medium (pt_libm.h: expD / logD / erfD; pt_scene.h: sqrtD)
__global__ void k_debug_libmd(int fn, const double *x, double *y, uint32_t n)
{
    const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    y[i] = fn == TGHIP_LIBM_EXPD ? ptlibm::expD(v) : fn == TGHIP_LIBM_LOGD ? ptlibm::logD(v) : fn == TGHIP_LIBM_ERFD ? ptlibm::erfD(v) : sqrtD(v);
}

int tghip_debug_libm(tghip_ctx *ctx, int fn, const float *x, float *y, size_t n)
{
    if (!ctx) return TGHIP_E_INVALID;
    if (n == 0) return TGHIP_OK;
    if (!x || !y || n > 0x3FFFFFFFu || fn < TGHIP_LIBM_SINF || fn > TGHIP_LIBM_SQRTD) { ctx->error = "invalid libm self-test arguments"; return TGHIP_E_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *dx = nullptr, *dy = nullptr;
    const bool dbl = fn >= TGHIP_LIBM_EXPD;                       // n doubles in, n doubles out
    const size_t nx = (fn == TGHIP_LIBM_ATAN2F || fn == TGHIP_LIBM_POWF || dbl) ? 2*n : n, ny = dbl ? 2*n : n;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&dx), nx*sizeof(float)));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dy), ny*sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, nx*sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        if (dbl) hipLaunchKernelGGL(k_debug_libmd, dim3(uint32_t((n + 255)/256)), dim3(256), 0, ctx->stream, fn, reinterpret_cast<const double *>(dx), reinterpret_cast<double *>(dy), uint32_t(n));
        else     hipLaunchKernelGGL(k_debug_libm, dim3(uint32_t((n + 255)/256)), dim3(256), 0, ctx->stream, fn, dx, dy, uint32_t(n));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(y, dy, ny*sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (e != hipSuccess) { ctx->error = hipGetErrorString(e); return TGHIP_E_HIP; }
    return TGHIP_OK;
}

extern "C++" void tghipDestroyRankComm(void *comm)
{
    std::lock_guard<std::mutex> lock(g_rccl.mutex);
    if (comm && g_rccl.load()) (void)g_rccl.commDestroy(static_cast<ncclComm_t>(comm));
}

// ---- the same reduce between PROCESSES (one process per GPU, SURVEY.md 8e: torch.distributed.run, mpirun, ...) ----
int tghip_comm_unique_id(void *id, size_t bytes)
{
    static_assert(sizeof(ncclUniqueId) == TGHIP_COMM_ID_BYTES, "TGHIP_COMM_ID_BYTES");
    if (!id || bytes < sizeof(ncclUniqueId)) return TGHIP_E_INVALID;
    std::lock_guard<std::mutex> lock(g_rccl.mutex);
    if (!g_rccl.load()) return TGHIP_E_UNSUPPORTED;
    ncclUniqueId u;
    if (g_rccl.getUniqueId(&u) != ncclSuccess) return TGHIP_E_HIP;
    std::memcpy(id, &u, sizeof(u));
    return TGHIP_OK;
}

int tghip_comm_init_rank(tghip_ctx *ctx, const void *id, size_t bytes, int nranks, int rank)
{
    if (!ctx || !id || bytes < sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks) return TGHIP_E_INVALID;
    if (ctx->failReduce) { ctx->error = "tghip_comm_init_rank: forced failure (the \"fail_reduce\" option)"; return TGHIP_E_HIP; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lock(g_rccl.mutex);
    if (!g_rccl.load()) { ctx->error = "tghip_comm_init_rank: librccl.so could not be loaded"; return TGHIP_E_UNSUPPORTED; }
    if (ctx->rankComm) { (void)g_rccl.commDestroy(static_cast<ncclComm_t>(ctx->rankComm)); ctx->rankComm = nullptr; }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.commInitRank(&comm, nranks, u, rank);
    if (r != ncclSuccess) { ctx->error = std::string("ncclCommInitRank: ") + g_rccl.errorString(r); return TGHIP_E_HIP; }
    ctx->rankComm = comm; ctx->rankCount = nranks; ctx->rankIndex = rank;
    return TGHIP_OK;
}

int tghip_reduce_framebuffer_rank(tghip_ctx *ctx, int root, float *rgb_sum, uint32_t *count, size_t npixels)
{
    if (!ctx) return TGHIP_E_INVALID;
    if (!ctx->haveScene) return TGHIP_E_NOSCENE;
    if (!ctx->rankComm) { ctx->error = "tghip_reduce_framebuffer_rank: no communicator (tghip_comm_init_rank)"; return TGHIP_E_INVALID; }
    if (root < 0 || root >= ctx->rankCount || npixels != size_t(ctx->width)*ctx->height) { ctx->error = "tghip_reduce_framebuffer_rank: root / pixel count mismatch"; return TGHIP_E_INVALID; }
    if (ctx->failReduce) { ctx->error = "tghip_reduce_framebuffer_rank: forced failure (the \"fail_reduce\" option)"; return TGHIP_E_HIP; }
    int w = tghip_wait(ctx);
    if (w != TGHIP_OK && w != TGHIP_E_ABORTED) return w;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool isRoot = ctx->rankIndex == root;
    if (isRoot && ctx->redCap < npixels) {
        if (ctx->redSum) (void)hipFree(ctx->redSum);
        if (ctx->redCount) (void)hipFree(ctx->redCount);
        ctx->redSum = nullptr; ctx->redCount = nullptr; ctx->redCap = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->redSum), npixels*3*sizeof(float)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->redCount), npixels*sizeof(uint32_t)));
        ctx->redCap = npixels;
    }
    const float *sum = ctx->extSum ? ctx->extSum : ctx->fbSum;
    const uint32_t *cnt = ctx->extCount ? ctx->extCount : ctx->fbCount;
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->rankComm);
    ncclResult_t r;
    {
        std::lock_guard<std::mutex> lock(g_rccl.mutex);
        r = g_rccl.groupStart();
        if (r == ncclSuccess) r = g_rccl.reduce(sum, isRoot ? ctx->redSum : nullptr, npixels*3, ncclFloat32, ncclSum, root, comm, ctx->stream);
        if (r == ncclSuccess) r = g_rccl.reduce(cnt, isRoot ? ctx->redCount : nullptr, npixels, ncclUint32, ncclSum, root, comm, ctx->stream);
        const ncclResult_t e = g_rccl.groupEnd();
        if (r == ncclSuccess) r = e;
    }
    if (r != ncclSuccess) { ctx->error = std::string("ncclReduce: ") + g_rccl.errorString(r); return TGHIP_E_HIP; }
    if (isRoot && rgb_sum) HIP_TRY(ctx, hipMemcpyAsync(rgb_sum, ctx->redSum, npixels*3*sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (isRoot && count) HIP_TRY(ctx, hipMemcpyAsync(count, ctx->redCount, npixels*sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
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
        // (closest hits of a scene with `instances` primitives are a matter of the reference's visiting order: the BVH2 walk, never the wide one)
        if (useWide(ctx) && !ctx->haveInstances) {
            const size_t ldsWide = size_t(std::max(ctx->wideDepth, 1))*256u*sizeof(uint2);
#define RAYS_WIDE(C, I) hipLaunchKernelGGL((k_trace_rays<C, false, I, true>), dim3(grid), dim3(256), ldsWide, ctx->stream, ctx->scene, dRays, dHits, uint32_t(n), ctx->pool.stats)
            if (ctx->haveInstances) { if (cnt) RAYS_WIDE(true, 1); else RAYS_WIDE(false, 1); }
            else                    { if (cnt) RAYS_WIDE(true, 0); else RAYS_WIDE(false, 0); }
#undef RAYS_WIDE
        }
        else if (ctx->haveInstances) {
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
    std::memset(ctx->walkStats, 0, sizeof(ctx->walkStats));
    return TGHIP_OK;
}

int tghip_get_walk_stats(tghip_ctx *ctx, int walk, uint64_t *out, int n)
{
    if (!ctx || !out || walk < 0 || walk > 1 || n < 0) return TGHIP_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = foldCounters(ctx);
    if (rc != TGHIP_OK) return rc;
    const int m = std::min(n, int(PT_WALK_STATS));
    for (int i = 0; i < m; ++i) out[i] = ctx->walkStats[walk][i];
    return m;
}

} // extern "C"
