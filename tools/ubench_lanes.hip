// Development aid (not product code): what SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU mean on gfx950, so that "active lanes per issued VALU
// instruction" (tools/pmc_variants.py --groups lane, bench.py roofline.valu.lane_utilisation) is read off a calibrated ratio and not off the
// counters' one-line descriptions.  One kernel per (instruction class, enabled lanes): a wave runs the same unrolled chain with EXEC = the
// first N lanes, N = 64, 48, 32, 16, 8, 1.  Run under `rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES`:
// the ratio THREAD_CYCLES / ACTIVE_INST of the N-lane kernel over the 64-lane kernel's is the N/64 the counter pair reports.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lanes.hip -o tools/bin/ubench_lanes && tools/bin/ubench_lanes
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define A_FMA(k)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_CVT(k)    asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k]));
#define A_RCP(k)    asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));

// KIND 0: v_fma_f32 (2 clocks per wave64 instruction), 1: v_cvt_f32_ubyte1 (3.2), 2: v_rcp_f32 (6.2)   [profiles/r5_ubench_valu.txt]
template<int KIND, int LANES>
__global__ __launch_bounds__(256) void k_lanes(int iters, uint32_t seed, uint32_t *out)
{
    uint32_t a[8];
    for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x*(k + 1);
    uint32_t b = seed | 1u, c = seed ^ 0x3f800000u;
    if ((threadIdx.x & 63u) < (uint32_t)LANES) {
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) { BODY4(A_FMA) } else if (KIND == 1) { BODY4(A_CVT) } else { BODY4(A_RCP) }
        }
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r ^= a[k];
    if (r == 0x12345u) out[0] = r;
}

template<int KIND, int LANES>
static int run(const char *name, uint32_t *out)
{
    const int iters = 2000, blocks = 256*8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_lanes<KIND, LANES>), dim3(blocks), dim3(256), 0, 0, 10, 1u, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_lanes<KIND, LANES>), dim3(blocks), dim3(256), 0, 0, iters, 1u, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waveInsts = (double)blocks*4.0*iters*32.0;
    std::printf("%-8s lanes %2d  %8.3f ms  %7.1f G wave-inst/s\n", name, LANES, ms, waveInsts/ms*1e-6);
    return 0;
}

// the same chain under an arbitrary EXEC mask (a kernel argument): WHICH lanes are enabled, not only how many -- the first run showed a cliff
// between 16 and 8 enabled lanes (fma 840 -> 233 G wave-instructions/s), which a walk's one- and two-lane divergent regions would sit on
__global__ __launch_bounds__(256) void k_mask(int iters, uint32_t seed, uint32_t *out, unsigned long long mask, int waves)
{
    uint32_t a[8];
    for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x*(k + 1);
    uint32_t b = seed | 1u, c = seed ^ 0x3f800000u;
    if (((mask >> (threadIdx.x & 63u)) & 1ull) && (int)(threadIdx.x >> 6) < waves) {
        for (int i = 0; i < iters; ++i) { BODY4(A_FMA) }
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r ^= a[k];
    if (r == 0x12345u) out[0] = r;
}
// MIXED: is the price of a narrow instruction paid per instruction, or only by code that runs narrow for long?  Every turn of the loop issues 32 fmas
// with every lane enabled and then 32 with `mask`; and a second form in which the waves of a SIMD differ -- odd waves of the workgroup run wide only.
__global__ __launch_bounds__(256) void k_mixed(int iters, uint32_t seed, uint32_t *out, unsigned long long mask, int oddWavesWideOnly)
{
    uint32_t a[8];
    for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x*(k + 1);
    uint32_t b = seed | 1u, c = seed ^ 0x3f800000u;
    const bool narrow = ((mask >> (threadIdx.x & 63u)) & 1ull) && !(oddWavesWideOnly && ((threadIdx.x >> 6) & 1u));
    for (int i = 0; i < iters; ++i) {
        BODY4(A_FMA)
        if (narrow) { BODY4(A_FMA) }
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r ^= a[k];
    if (r == 0x12345u) out[0] = r;
}
static int runMixed(const char *name, unsigned long long mask, int oddWide, uint32_t *out)
{
    const int iters = 2000, blocks = 256*8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(256), 0, 0, 10, 1u, out, mask, oddWide);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(256), 0, 0, iters, 1u, out, mask, oddWide);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("mixed %-40s %016llx (%2d lanes) odd waves wide only %d  %8.3f ms  (32 wide + 32 masked fmas per turn; 32 wide alone: see `all` with half the instructions)\n",
                name, mask, __builtin_popcountll(mask), oddWide, ms);
    return 0;
}

static int runMask(const char *name, unsigned long long mask, int waves, int blocksPerCu, uint32_t *out)
{
    const int iters = 2000, blocks = 256*blocksPerCu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mask, dim3(blocks), dim3(256), 0, 0, 10, 1u, out, mask, waves);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mask, dim3(blocks), dim3(256), 0, 0, iters, 1u, out, mask, waves);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waveInsts = (double)blocks*waves*iters*32.0;
    std::printf("mask %-34s %016llx (%2d lanes) %d waves/block x %d blocks/CU %8.3f ms  %7.1f G wave-inst/s\n", name, mask, __builtin_popcountll(mask), waves, blocksPerCu, ms, waveInsts/ms*1e-6);
    return 0;
}

int main(int argc, char **argv)
{
    uint32_t *out = nullptr;
    CHECK(hipMalloc(&out, 64));
    if (argc > 1 && std::string(argv[1]) == "masks") {
        struct { const char *name; unsigned long long m; } pats[] = {
            {"all", ~0ull}, {"first 16", 0xFFFFull}, {"first 12", 0xFFFull}, {"first 10", 0x3FFull}, {"first 9", 0x1FFull}, {"first 8", 0xFFull},
            {"first 4", 0xFull}, {"first 2", 0x3ull}, {"first 1", 0x1ull}, {"lanes 32..39", 0xFFull << 32}, {"lanes 56..63", 0xFFull << 56},
            {"every 8th (8 lanes)", 0x0101010101010101ull}, {"every 4th (16 lanes)", 0x1111111111111111ull}, {"every 16th (4 lanes)", 0x0001000100010001ull},
            {"4 + 4 in two halves", 0x0000000F0000000Full}, {"one lane per 16 + first 8", 0x00010001000100FFull}, {"lanes 0..7 and 32..39", 0x000000FF000000FFull},
        };
        for (auto &p : pats) if (runMask(p.name, p.m, 4, 8, out)) return 1;
        // fewer waves per SIMD: is the cliff an issue-rate effect that more waves hide, or a per-instruction cost?
        for (int bpc : {1, 2, 4}) { if (runMask("all", ~0ull, 4, bpc, out) || runMask("first 8", 0xFFull, 4, bpc, out) || runMask("first 1", 1ull, 4, bpc, out)) return 1; }
        for (int odd = 0; odd < 2; ++odd)
            if (runMixed("wide + all", ~0ull, odd, out) || runMixed("wide + first 16", 0xFFFFull, odd, out) || runMixed("wide + first 8", 0xFFull, odd, out) ||
                runMixed("wide + first 1", 1ull, odd, out) || runMixed("wide + none", 0ull, odd, out)) return 1;
        CHECK(hipFree(out));
        return 0;
    }
#define RUN_ALL(KIND, NAME) \
    if (run<KIND, 64>(NAME, out) || run<KIND, 48>(NAME, out) || run<KIND, 32>(NAME, out) || run<KIND, 16>(NAME, out) || run<KIND, 8>(NAME, out) || run<KIND, 1>(NAME, out)) return 1;
    RUN_ALL(0, "fma")
    RUN_ALL(1, "cvt_ub")
    RUN_ALL(2, "rcp")
    CHECK(hipFree(out));
    return 0;
}
