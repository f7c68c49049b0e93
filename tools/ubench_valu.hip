// Development aid (not product code): issue cost of the VALU instructions the traversal kernels are made of, relative to v_fma_f32.
// The loop's VALU line in bench.py prices every wave instruction at 2 cycles (MI355X_MICROARCH.md: v_fma_f32, wave64); the walks are
// mostly integer / compare / select / convert instructions -- this measures what THOSE cost per SIMD, with 8 waves per SIMD and 8
// independent chains per wave, so that neither latency nor dependencies bound the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/bin/ubench_valu && tools/bin/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(X) REP8(X) REP8(X) REP8(X) REP8(X)

// one kernel per instruction: a[k] are eight independent accumulators, b / c loop-invariant operands
#define DEFINE_KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(int iters, uint32_t seed, uint32_t *out)                      \
    {                                                                                                             \
        uint32_t a[8];                                                                                            \
        for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x*(k + 1);                                            \
        uint32_t b = seed | 1u, c = seed ^ 0x3f800000u;                                                            \
        for (int i = 0; i < iters; ++i) {                                                                         \
            BODY4(ASM)                                                                                            \
        }                                                                                                         \
        uint32_t r = 0;                                                                                           \
        for (int k = 0; k < 8; ++k) r ^= a[k];                                                                    \
        if (r == 0x12345u) out[0] = r;                                                                            \
    }

#define A_FMA(k)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_MUL(k)      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_ADD(k)      asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MAX(k)      asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MAX3(k)     asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_AND(k)      asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_ADDU(k)     asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_LSHL(k)     asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k]));
#define A_LSHLADD(k)  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
#define A_BFE(k)      asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[k]));
#define A_AND_OR(k)   asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_CVT_UB(k)   asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k]));
#define A_CVT_U32(k)  asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[k]));
#define A_CNDMASK(k)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));
#define A_CMP(k)      asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define A_CMP_CND(k)  asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
#define A_MUL_LO(k)   asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MAD_U24(k)  asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_RCP(k)      asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define A_SQRT(k)     asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define A_BCNT(k)     asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_FFBL(k)     asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[k]));
#define A_MOV(k)      asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b));
#define A_XOR3(k)     asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_PERM(k)     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_FMAC(k)     asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));

#define A_FMAMIX(k)   asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_FMAMIXH(k)  asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_CVT_F16(k)  asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[k]));
#define A_MED3(k)     asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_MIN3(k)     asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_LSHLOR(k)   asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b));
#define A_OR3(k)      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_SUB(k)      asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_PKMUL(k)    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2[k]) : "v"(b2));
// round 6: the opcodes tools/isa_histogram.py found in the kernels that rounds 1-5 had not timed (the division sequence, SGPR-spill lane moves, ...)
#define A_DIVSCALE(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(c) : "vcc");
#define A_DIVFMAS(k)  asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");
#define A_DIVFIXUP(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define A_READLANE(k) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg[k]) : "v"(a[k]));
#define A_WRITELANE(k) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(a[k]) : "s"(sg[k]));
#define A_XOR(k)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_LSHR(k)     asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[k]));
#define A_BFEI(k)     asm volatile("v_bfe_i32 %0, %0, 3, 8" : "+v"(a[k]));
#define A_CVT_I32(k)  asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
#define A_MIN(k)      asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_CMPU(k)     asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define A_RSQ(k)      asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
#define A_LDEXP(k)    asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define A_MULHI(k)    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
DEFINE_KERNEL(divscale, A_DIVSCALE)
DEFINE_KERNEL(divfmas, A_DIVFMAS)
DEFINE_KERNEL(divfixup, A_DIVFIXUP)
DEFINE_KERNEL(xor_, A_XOR)
DEFINE_KERNEL(lshr, A_LSHR)
DEFINE_KERNEL(bfei, A_BFEI)
DEFINE_KERNEL(cvt_i32, A_CVT_I32)
DEFINE_KERNEL(min_, A_MIN)
DEFINE_KERNEL(cmpu, A_CMPU)
DEFINE_KERNEL(rsq, A_RSQ)
DEFINE_KERNEL(ldexp_, A_LDEXP)
DEFINE_KERNEL(mulhi, A_MULHI)
__global__ __launch_bounds__(256) void k_readlane(int iters, uint32_t seed, uint32_t *out)
{
    uint32_t a[8], sg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x*(k + 1);
    for (int i = 0; i < iters; ++i) { BODY4(A_READLANE) }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r ^= a[k] ^ sg[k];
    if (r == 0x12345u) out[0] = r;
}
__global__ __launch_bounds__(256) void k_writelane(int iters, uint32_t seed, uint32_t *out)
{
    uint32_t a[8], sg[8];
    for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x*(k + 1); sg[k] = seed + k; }
    for (int i = 0; i < iters; ++i) { BODY4(A_WRITELANE) }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r ^= a[k];
    if (r == 0x12345u) out[0] = r;
}
// f64 add / mul (Phong's pow(double) and the libm restatements' double arithmetic)
__global__ __launch_bounds__(256) void k_add64(int iters, uint32_t seed, uint32_t *out)
{
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = double(seed + threadIdx.x*(k + 1));
    double b = 1.0000001;
    for (int i = 0; i < iters; ++i) {
#define A_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        BODY4(A_ADD64)
    }
    double r = 0.0;
    for (int k = 0; k < 8; ++k) r += a[k];
    if (r == 12345.0) out[0] = 1u;
}
__global__ __launch_bounds__(256) void k_mul64(int iters, uint32_t seed, uint32_t *out)
{
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = double(seed + threadIdx.x*(k + 1));
    double b = 1.0000001;
    for (int i = 0; i < iters; ++i) {
#define A_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        BODY4(A_MUL64)
    }
    double r = 0.0;
    for (int k = 0; k < 8; ++k) r += a[k];
    if (r == 12345.0) out[0] = 1u;
}
DEFINE_KERNEL(fmamix, A_FMAMIX)
DEFINE_KERNEL(fmamixh, A_FMAMIXH)
DEFINE_KERNEL(cvt_f16, A_CVT_F16)
DEFINE_KERNEL(med3, A_MED3)
DEFINE_KERNEL(min3, A_MIN3)
DEFINE_KERNEL(lshlor, A_LSHLOR)
DEFINE_KERNEL(or3, A_OR3)
DEFINE_KERNEL(sub, A_SUB)
DEFINE_KERNEL(fma, A_FMA)
DEFINE_KERNEL(mul, A_MUL)
DEFINE_KERNEL(add, A_ADD)
DEFINE_KERNEL(max, A_MAX)
DEFINE_KERNEL(max3, A_MAX3)
DEFINE_KERNEL(and_, A_AND)
DEFINE_KERNEL(addu, A_ADDU)
DEFINE_KERNEL(lshl, A_LSHL)
DEFINE_KERNEL(lshladd, A_LSHLADD)
DEFINE_KERNEL(bfe, A_BFE)
DEFINE_KERNEL(and_or, A_AND_OR)
DEFINE_KERNEL(cvt_ub, A_CVT_UB)
DEFINE_KERNEL(cvt_u32, A_CVT_U32)
DEFINE_KERNEL(cndmask, A_CNDMASK)
DEFINE_KERNEL(cmp, A_CMP)
DEFINE_KERNEL(cmp_cnd, A_CMP_CND)
DEFINE_KERNEL(mul_lo, A_MUL_LO)
DEFINE_KERNEL(mad_u24, A_MAD_U24)
DEFINE_KERNEL(rcp, A_RCP)
DEFINE_KERNEL(sqrt_, A_SQRT)
DEFINE_KERNEL(bcnt, A_BCNT)
DEFINE_KERNEL(ffbl, A_FFBL)
DEFINE_KERNEL(mov, A_MOV)
DEFINE_KERNEL(xor3, A_XOR3)
DEFINE_KERNEL(perm, A_PERM)
DEFINE_KERNEL(fmac, A_FMAC)

// packed f32 (two lanes' worth of work per instruction): 64-bit operands
__global__ __launch_bounds__(256) void k_pk_fma(int iters, uint32_t seed, uint32_t *out)
{
    typedef float F2 __attribute__((ext_vector_type(2)));
    F2 a[8];
    for (int k = 0; k < 8; ++k) a[k] = F2{float(seed + threadIdx.x*(k + 1)), float(k)};
    F2 b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    for (int i = 0; i < iters; ++i) {
#define A_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        BODY4(A_PKFMA)
    }
    float r = 0.0f;
    for (int k = 0; k < 8; ++k) r += a[k].x + a[k].y;
    if (r == 12345.0f) out[0] = 1u;
}
// f64 fma
__global__ __launch_bounds__(256) void k_fma64(int iters, uint32_t seed, uint32_t *out)
{
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = double(seed + threadIdx.x*(k + 1));
    double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#define A_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        BODY4(A_FMA64)
    }
    double r = 0.0;
    for (int k = 0; k < 8; ++k) r += a[k];
    if (r == 12345.0) out[0] = 1u;
}

typedef void (*Kernel)(int, uint32_t, uint32_t *);
struct Entry { const char *name; Kernel fn; int perAsm; };

int main()
{
    const Entry entries[] = {
        {"v_fma_f32", k_fma, 1}, {"v_fmac_f32", k_fmac, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_max_f32", k_max, 1}, {"v_max3_f32", k_max3, 1},
        {"v_pk_fma_f32", k_pk_fma, 1}, {"v_fma_mix_f32 (f16 lo)", k_fmamix, 1}, {"v_fma_mix_f32 (f16 hi)", k_fmamixh, 1}, {"v_cvt_f32_f16", k_cvt_f16, 1},
        {"v_med3_f32", k_med3, 1}, {"v_min3_f32", k_min3, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_or3_b32", k_or3, 1}, {"v_sub_f32", k_sub, 1}, {"v_fma_f64", k_fma64, 1},
        {"v_and_b32", k_and_, 1}, {"v_add_u32", k_addu, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_and_or_b32", k_and_or, 1}, {"v_xad_u32", k_xor3, 1}, {"v_perm_b32", k_perm, 1}, {"v_mov_b32", k_mov, 1},
        {"v_cvt_f32_ubyte1", k_cvt_ub, 1}, {"v_cvt_f32_u32", k_cvt_u32, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp_lt_f32", k_cmp, 1}, {"v_cmp + v_cndmask", k_cmp_cnd, 2},
        {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_ffbl_b32", k_ffbl, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt_, 1},
        {"v_div_scale_f32", k_divscale, 1}, {"v_div_fmas_f32", k_divfmas, 1}, {"v_div_fixup_f32", k_divfixup, 1}, {"v_readlane_b32", k_readlane, 1},
        {"v_writelane_b32", k_writelane, 1}, {"v_xor_b32", k_xor_, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_bfe_i32", k_bfei, 1}, {"v_cvt_i32_f32", k_cvt_i32, 1},
        {"v_min_f32", k_min_, 1}, {"v_cmp_eq_u32", k_cmpu, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_ldexp_f32", k_ldexp_, 1}, {"v_mul_hi_u32", k_mulhi, 1},
        {"v_add_f64", k_add64, 1}, {"v_mul_f64", k_mul64, 1},
    };
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out = nullptr;
    CHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4096;
    const int blocksPerCu = 8;                   // 8 x 256 threads = 32 waves per CU = 8 per SIMD
    double fmaNs = 0.0;
    std::printf("%d CUs, %d blocks of 256 threads per CU, %d x 32 instructions per wave\n", cus, blocksPerCu, iters);
    std::printf("%-22s %10s %12s %14s\n", "instruction", "ms", "rel. to fma", "cycles/SIMD*");
    for (const Entry &e : entries) {
        hipLaunchKernelGGL(e.fn, dim3(cus*blocksPerCu), dim3(256), 0, 0, 64, 1u, out);      // warm-up
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.fn, dim3(cus*blocksPerCu), dim3(256), 0, 0, iters, 1u, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double perInstrNs = double(best)*1e6/(double(iters)*32.0*e.perAsm*8.0);         // per wave instruction per SIMD (8 waves share it)
        if (fmaNs == 0.0) fmaNs = perInstrNs;
        std::printf("%-22s %10.3f %12.2f %14.2f\n", e.name, best, perInstrNs/fmaNs, 2.0*perInstrNs/fmaNs);
    }
    std::printf("* with v_fma_f32 = 2 cycles per wave64 instruction (MI355X_MICROARCH.md)\n");
    return 0;
}
