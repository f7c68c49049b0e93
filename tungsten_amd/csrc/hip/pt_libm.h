// The host libm's sinf / cosf / logf / expf on the device, bit for bit.
//
// The reference calls std::sin / std::cos / std::log / std::exp on floats wherever it turns random numbers into directions and
// distances (SampleWarp.hpp:42-57, Microfacet.hpp:100-116, HomogeneousMedium.cpp, ...), i.e. glibc's sinf / cosf / logf / expf.
// ocml's versions differ from those in the last bit on a fifth of the arguments -- one ulp in a bounce direction, invisible in a
// pixel, but a path is a chaotic function of its hits: the fork shows up as the per-sample divergence DESIGN.md section 7 tabulates,
// and where a quantity is ill-conditioned (chooseLight's weights for millimetre-sized emitters) in per cents of the samples.
//
// glibc 2.35 (the image's) computes all four in DOUBLE precision with short polynomials -- the "optimized routines" algorithms
// (sysdeps/ieee754/flt-32/s_sincosf.h, e_logf.c, e_expf.c) -- and on an x86-64 host with FMA3 runs the variants compiled with
// contraction.  The functions below restate those algorithms with the fused operations spelt out (the kernels are compiled with
// -ffp-contract=off, so exactly these are fused).  They were matched against the image's libm EXHAUSTIVELY on the host: every float
// in [0, 120) for sinf / cosf, every positive float for logf, every float in (-88, 88) for expf -- zero mismatches
// (tests/test_host.py::test_libm_restatements_match_the_host_libm runs a sample of that; oracle/libm_host.cpp is this header
// compiled for the host).  Outside those ranges (|x| >= 120: glibc's Payne-Hanek reduction; |x| >= 88) they return ocml's value;
// no call site gets there (phi = 2 pi xi, theta = pi v, -sigma t of a surviving path).
//
// Plain C++: no HIP header, so that the same text compiles for the host test.
#ifndef TGAMD_PT_LIBM_H_
#define TGAMD_PT_LIBM_H_

#include <stdint.h>

#ifndef PT_LIBM_FN
#define PT_LIBM_FN __device__ __forceinline__
#define PT_LIBM_TABLE __device__ const
#endif

namespace ptlibm {

PT_LIBM_FN uint32_t f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
PT_LIBM_FN float u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
PT_LIBM_FN uint64_t d2u(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
PT_LIBM_FN double u2d(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
PT_LIBM_FN uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ffu; }

// ---- sinf / cosf: s_sincosf.h (reduce_fast without TOINT_INTRINSICS, sinf_poly), s_sinf.c, s_cosf.c ----
PT_LIBM_FN double sinPoly(double x, double x2)       // sine on [-pi/4, pi/4]
{
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double x3 = x*x2;
    const double s1 = __builtin_fma(x2, S3, S2);
    const double x7 = x3*x2;
    const double s = __builtin_fma(x3, S1, x);
    return __builtin_fma(x7, s1, s);
}
PT_LIBM_FN double cosPoly(double x2)                 // cosine on [-pi/4, pi/4]
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double x4 = x2*x2;
    const double c2 = __builtin_fma(x2, C4, C3);
    const double c1 = __builtin_fma(x2, C1, C0);
    const double x6 = x4*x2;
    const double c = __builtin_fma(x4, C2, c1);
    return __builtin_fma(x6, c2, c);
}
// x = y - n pi/2 with the quadrant n in bits 24.. of y*(2/pi)*2^24, rounded by adding half (reduce_fast)
PT_LIBM_FN double reduceFast(double x, int &n)
{
    const double hpiInv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x*hpiInv;
    n = ((int32_t)r + 0x800000) >> 24;
    return __builtin_fma(-(double)n, hpi, x);
}
// (quadrant n, reduced x) -> sin: even quadrants the sine polynomial with the sign of quadrants 1 and 2 on x, odd ones the cosine
// polynomial, negated in quadrants 2 and 3 (the second __sincosf_table entry)
PT_LIBM_FN float sinQuadrant(double x, int n)
{
    if ((n & 1) == 0)
        return (float)sinPoly(((n + 1) & 2) ? -x : x, x*x);
    const double c = cosPoly(x*x);
    return (float)((n & 2) ? -c : c);
}
PT_LIBM_FN bool sincosInRange(float y) { return abstop12(y) < abstop12(120.0f); }
PT_LIBM_FN float sinfCore(float y)                  // |y| < 120
{
    const double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f))
            return y;
        return (float)sinPoly(x, x*x);
    }
    int n;
    const double r = reduceFast(x, n);
    return sinQuadrant(r, n);
}
PT_LIBM_FN float cosfCore(float y)                  // |y| < 120: cos y = sin(y + pi/2), one quadrant on
{
    const double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f))
            return 1.0f;
        return (float)cosPoly(x*x);
    }
    int n;
    const double r = reduceFast(x, n);
    return sinQuadrant(r, n + 1);
}
// both at once for the call sites that need the pair (one reduction)
PT_LIBM_FN void sincosfCore(float y, float &s, float &c)
{
    const double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) { s = y; c = 1.0f; return; }
        const double x2 = x*x;
        s = (float)sinPoly(x, x2);
        c = (float)cosPoly(x2);
        return;
    }
    int n;
    const double r = reduceFast(x, n);
    // both polynomials once: sinPoly is odd operation by operation, so the sign sinQuadrant puts on its argument can go on the result
    const double r2 = r*r, sp = sinPoly(r, r2), cp = cosPoly(r2);
    const int m = n + 1;
    s = (n & 1) == 0 ? (float)(((n + 1) & 2) ? -sp : sp) : (float)((n & 2) ? -cp : cp);
    c = (m & 1) == 0 ? (float)(((m + 1) & 2) ? -sp : sp) : (float)((m & 2) ? -cp : cp);
}

// ---- logf: e_logf.c with __logf_data (16 intervals of [sqrt(2)/2, sqrt(2)), degree-3 polynomial) ----
PT_LIBM_TABLE double g_logfTable[16][2] = {     // {1/c, log c} of the interval centres
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
    {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
    {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
    {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
PT_LIBM_FN bool logInRange(float x) { const uint32_t ix = f2u(x); return ix - 0x00800000u < 0x7f800000u - 0x00800000u; }   // positive, normal, finite
PT_LIBM_FN float logfCore(float x)                  // x positive, normal and finite
{
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const uint32_t ix = f2u(x);
    if (ix == 0x3f800000u)
        return 0.0f;
    // x = 2^k z with z in [0x3f330000, 2 x that), split into 16 intervals
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = g_logfTable[i][0], logc = g_logfTable[i][1];
    const double z = (double)u2f(iz);
    // log(x) = log1p(z/c - 1) + log(c) + k ln 2
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)k, Ln2, logc);
    const double r2 = r*r;
    double y = __builtin_fma(A1, r, A2);
    y = __builtin_fma(A0, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return (float)y;
}

// ---- expf: e_expf.c with __exp2f_data (N = 32: exp(x) = 2^(k/32) 2^(r/32), degree-3 polynomial) ----
PT_LIBM_TABLE uint64_t g_exp2fTable[32] = {     // bits of 2^(i/32) minus i << 47
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
PT_LIBM_FN bool expInRange(float x) { return abstop12(x) < abstop12(88.0f); }
PT_LIBM_FN float expfCore(float x)                  // |x| < 88
{
    const double N = 32.0, Shift = 0x1.8p+52, InvLn2N = 0x1.71547652b82fep+0*N;
    const double C0 = 0x1.c6af84b912394p-5/N/N/N, C1 = 0x1.ebfce50fac4f3p-3/N/N, C2 = 0x1.62e42ff0c52d6p-1/N;
    const double xd = x;
    // x N/ln 2 = k + r, r in [-1/2, 1/2]: k by adding and subtracting 1.5 x 2^52
    double z = InvLn2N*xd;
    double kd = z + Shift;
    const uint64_t ki = d2u(kd);
    kd -= Shift;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    const double s = u2d(g_exp2fTable[ki & 31u] + (ki << 47));
    z = __builtin_fma(C0, r, C1);
    const double r2 = r*r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    return (float)(y*s);
}

}  // namespace ptlibm

#endif
