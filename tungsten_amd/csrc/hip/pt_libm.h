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
// -ffp-contract=off, so exactly these are fused).  They were matched against the image's libm EXHAUSTIVELY on the host: every float32 bit
// pattern of either sign -- for sinf / cosf those with |x| < 120, for logf / expf ALL of them, special cases included -- with zero
// mismatches (tools/libm_sweep.py; tests/test_host.py::test_libm_restatements_match_the_host_libm runs every fifth; oracle/libm_host.cpp is
// this header compiled for the host), and the device's results are held to the host libm by tests/test_gpu_libm.py.  Outside those ranges
// for sin / cos -- which no call site reaches (phi = 2 pi xi, theta = pi v) -- pt_math.h's wrappers fold the angle into [-pi, pi] in double.
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
PT_LIBM_FN bool sincosInRange(float y) { return abstop12(y) < abstop12(120.0f); }
// sin y and cos y, |y| < 120, from ONE quadrant reduction and both polynomials, without a branch.  glibc's code has three: |y| < 2^-12
// (returns y and 1), |y| < pi/4 (no reduction) and the general one with its four quadrants.  They coincide with what is computed here:
// below pi/4 the reduction finds n = 0 and returns y itself (fma(-0, pi/2, y)); below 2^-12 the polynomials round to y and to 1; and the
// sign glibc puts on the sine polynomial's ARGUMENT in quadrants 1 and 2 can go on its result, the polynomial being odd operation by
// operation.  (Checked like everything in this header: every float, both signs, against the host libm.)  On the device that is 21 double
// operations per call for every lane instead of two or three divergent paths of the same length.
//   sin: quadrant 0: sp, 1: cp, 2: -sp, 3: -cp        cos = sin one quadrant on: 0: cp, 1: -sp, 2: -cp, 3: sp
PT_LIBM_FN void sincosfCore(float y, float &s, float &c)
{
    int n;
    const double r = reduceFast((double)y, n);
    const double r2 = r*r, sp = sinPoly(r, r2), cp = cosPoly(r2);
    const double sv = (n & 1) ? cp : sp, cv = (n & 1) ? sp : cp;
    s = (float)((n & 2) ? -sv : sv);
    c = (float)(((n + 1) & 2) ? -cv : cv);
    if (f2u(y) == 0x80000000u) s = y;              // sin(-0) = -0: the one float for which the polynomial's sum (+0 + -0) is not glibc's `return y`
}
PT_LIBM_FN float sinfCore(float y) { float s, c; sincosfCore(y, s, c); return s; }
PT_LIBM_FN float cosfCore(float y) { float s, c; sincosfCore(y, s, c); return c; }

// ---- logf: e_logf.c with __logf_data (16 intervals of [sqrt(2)/2, sqrt(2)), degree-3 polynomial) ----
PT_LIBM_TABLE double g_logfTable[16][2] = {     // {1/c, log c} of the interval centres
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
    {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
    {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
    {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
PT_LIBM_FN float logfCore(float x)                  // x positive, normal and finite (logfAll: every float)
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

// every float: e_logf.c's special cases in front of the core -- log(+-0) = -inf, log(negative) = NaN, inf and NaN through, subnormals scaled
// by 2^23 with the exponent corrected in the bit pattern
PT_LIBM_FN float logfAll(float x)
{
    uint32_t ix = f2u(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix*2u == 0u) return -__builtin_huge_valf();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix*2u >= 0xff000000u) return (x - x)/(x - x);
        ix = f2u(x*0x1p23f) - (23u << 23);
    }
    return logfCore(u2f(ix));
}

// ---- expf: e_expf.c with __exp2f_data (N = 32: exp(x) = 2^(k/32) 2^(r/32), degree-3 polynomial) ----
PT_LIBM_TABLE uint64_t g_exp2fTable[32] = {     // bits of 2^(i/32) minus i << 47
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
PT_LIBM_FN float expfCore(float x)                  // -103.97 < x < 88.72 (expfAll: every float)
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

// every float: e_expf.c's special cases in front of the core -- which itself covers (-103.97, 88.72): the double product rounds to the subnormal
// float below 2^-126 -- exp(-inf) = 0, NaN and +inf through, overflow to inf above log(2^128), underflow to 0 below log(2^-150)
PT_LIBM_FN float expfAll(float x)
{
    if (abstop12(x) >= abstop12(88.0f)) {
        if (f2u(x) == 0xff800000u) return 0.0f;
        if (abstop12(x) >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_huge_valf();
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    return expfCore(x);
}

// ---- atanf / atan2f: s_atanf.c / e_atan2f.c, fdlibm's float algorithm (argument reduction to one of four intervals, odd / even split of an
// 11-term polynomial), evaluated WITHOUT contraction: that is what the image's libm returns for every float (atanf) and for 4 x 10^8 random
// pairs (atan2f).  Called by the kernels since round 4 (pt_math.h: atan2fH / powfH / cbrtfH): the texture coordinates of environment maps and
// spheres, the Davis transmittances, the Rayleigh phase function.
PT_LIBM_FN float atanfCore(float x)
{
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f, a3 = -1.1111110449e-01f, a4 = 9.0908870101e-02f,
                a5 = -7.6918758452e-02f, a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f, a8 = 4.9768779427e-02f, a9 = -3.6531571299e-02f,
                a10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff;
    // One straight line for every lane (the four-interval branch nest of s_atanf.c cost the shading kernels 1 100 instructions per call
    // site): the reduced argument is ONE quotient num/den with the operands selected per interval -- below 7/16 it is x/1, exact, i.e. x
    // itself --, both result forms are evaluated and selected, and the special cases (|x| >= 2^25, NaN, |x| < 2^-29) override at the end.
    const float ax = __builtin_fabsf(x);
    const bool direct = ix < 0x3ee00000;                      // |x| < 7/16: no reduction
    const int iv = ix < 0x3f300000 ? 0 : ix < 0x3f980000 ? 1 : ix < 0x401c0000 ? 2 : 3;
    const float num = direct ? x : iv == 0 ? 2.0f*ax - 1.0f : iv == 1 ? ax - 1.0f : iv == 2 ? ax - 1.5f : -1.0f;
    const float den = direct ? 1.0f : iv == 0 ? 2.0f + ax : iv == 1 ? ax + 1.0f : iv == 2 ? 1.0f + 1.5f*ax : ax;
    const float hi = iv == 0 ? hi0 : iv == 1 ? hi1 : iv == 2 ? hi2 : hi3;
    const float lo = iv == 0 ? lo0 : iv == 1 ? lo1 : iv == 2 ? lo2 : lo3;
    const float t = num/den;
    const float z = t*t, w = z*z;
    const float s1 = z*(a0 + w*(a2 + w*(a4 + w*(a6 + w*(a8 + w*a10)))));
    const float s2 = w*(a1 + w*(a3 + w*(a5 + w*(a7 + w*a9))));
    const float ts = t*(s1 + s2);
    const float reduced = hi - ((ts - lo) - t);
    float r = direct ? t - ts : hx < 0 ? -reduced : reduced;
    if (ix < 0x31000000) r = x;                               // |x| < 2^-29
    if (ix >= 0x4c000000) r = ix > 0x7f800000 ? x + x : hx > 0 ? hi3 + lo3 : -hi3 - lo3;   // |x| >= 2^25, NaN
    return r;
}
PT_LIBM_FN float atan2fCore(float y, float x)
{
    const float tiny = 1.0e-30f, pio4 = 7.8539818525e-01f, pio2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, piLo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff, hy = (int32_t)f2u(y), iy = hy & 0x7fffffff;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);            // 2 sign(x) + sign(y)
    // the general case first, the special cases of e_atan2f.c as overrides in reverse order of their tests.  (x == 1 needs none: y/1 is y, and
    // atanfCore is odd operation by operation, so sign(y) atanf(|y|) IS atanf(y) -- checked with the rest, oracle/libm_host.cpp.)
    const int k = (iy - ix) >> 23;
    float z = atanfCore(__builtin_fabsf(y/x));
    if (hx < 0 && k < -60) z = 0.0f;
    if (k > 60) z = pio2 + 0.5f*piLo;
    float r = m == 0 ? z : m == 1 ? u2f(f2u(z) ^ 0x80000000u) : m == 2 ? pi - (z - piLo) : (z - piLo) - pi;
    if (iy == 0x7f800000) r = hy < 0 ? -pio2 - tiny : pio2 + tiny;
    if (ix == 0x7f800000)
        r = iy == 0x7f800000 ? (m == 0 ? pio4 + tiny : m == 1 ? -pio4 - tiny : m == 2 ? 3.0f*pio4 + tiny : -3.0f*pio4 - tiny)
                             : (m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) r = hy < 0 ? -pio2 - tiny : pio2 + tiny;
    if (iy == 0) r = m < 2 ? y : m == 2 ? pi + tiny : -pi - tiny;
    if (ix > 0x7f800000 || iy > 0x7f800000) r = x + y;
    return r;
}

// ---- powf: e_powf.c -- log2(x) by the 16-interval table of __powf_log2_data and a degree-5 polynomial, y log2(x) in double, 2^(...) by
// the exp2 table above; the FMA3 variant (every multiply-add fused).  x positive and normal, y finite, the result a normal float.
PT_LIBM_TABLE double g_powfLog2Table[16][2] = {     // {1/c, log2 c}
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},
    {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
    {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
    {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
PT_LIBM_FN bool powInRange(float x, float y)
{
    const uint32_t ix = f2u(x), iy = f2u(y);
    return ix - 0x00800000u < 0x7f800000u - 0x00800000u && (iy & 0x7fffffffu) < 0x7f800000u && (iy & 0x7fffffffu) != 0u;
}
// false when y log2(x) is outside (-126, 126): glibc's overflow / underflow paths, left to the caller's fallback
PT_LIBM_FN bool powfCore(float x, float y, float &result)
{
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const uint32_t ix = f2u(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const int k = (int32_t)top >> 23;
    const double invc = g_powfLog2Table[i][0], logc = g_powfLog2Table[i][1];
    const double z = (double)u2f(ix - top);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r*r;
    double yy = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2*r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);                                 // log2(x)
    const double ylogx = (double)y*yy;
    if (((d2u(ylogx) >> 47) & 0xffffu) >= (d2u(126.0) >> 47))
        return false;
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1, Shift = 0x1.8p+52/32.0;
    double kd = ylogx + Shift;
    const uint64_t ki = d2u(kd);
    kd -= Shift;
    const double rr = ylogx - kd;
    const double s = u2d(g_exp2fTable[ki & 31u] + (ki << 47));
    const double zz = __builtin_fma(C0, rr, C1);
    const double rr2 = rr*rr;
    double e = __builtin_fma(C2, rr, 1.0);
    e = __builtin_fma(zz, rr2, e);
    result = (float)(e*s);
    return true;
}

// ---- cbrtf: s_cbrtf.c -- frexp, a quadratic first guess and one Halley step in double, the cube root of the exponent's remainder from a
// five-entry table, ldexp.  (Insensitive to contraction: every fused and unfused variant gives glibc's result for every float.)
PT_LIBM_FN float cbrtfCore(float x)
{
    const uint32_t ux = f2u(x) & 0x7fffffffu;
    if (ux == 0u || ux >= 0x7f800000u)
        return x + x;
    // frexp: |x| = xm 2^xe, xm in [0.5, 1)
    int xe;
    float xm;
    if (ux < 0x00800000u) {                         // subnormal: normalise by 2^25 first
        const uint32_t un = f2u(u2f(ux)*0x1p25f);
        xe = (int)(un >> 23) - 126 - 25;
        xm = u2f((un & 0x007fffffu) | 0x3f000000u);
    } else {
        xe = (int)(ux >> 23) - 126;
        xm = u2f((ux & 0x007fffffu) | 0x3f000000u);
    }
    const double dxm = (double)xm;
    const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066*dxm)*dxm);
    const float t2 = u*u*u;
    const int rem = xe % 3;                         // C remainder: sign of xe
    const double factor = rem == -2 ? 1.0/1.5874010519681994748 : rem == -1 ? 1.0/1.2599210498948731648 : rem == 0 ? 1.0
                        : rem == 1 ? 1.2599210498948731648 : 1.5874010519681994748;
    const float ym = (float)((double)u*((double)t2 + 2.0*dxm)/(2.0*(double)t2 + dxm)*factor);
    // ldexp(ym, xe / 3): ym in [0.5, 1.6), xe / 3 in [-50, 43) -- the product is a normal float, so an exact scaling by a power of two
    const float scaled = ym*u2f((uint32_t)(xe/3 + 127) << 23);
    return (f2u(x) >> 31) ? -scaled : scaled;
}

// ---- tanf: s_tanf.c + k_tanf.c (glibc 2.35: fdlibm's float kernel, no FMA variant) behind the double-precision quadrant reduction of
// s_sincosf.h, head and tail handed to the kernel as two floats (read off the image's libm: `objdump -d libm.so.6`, tanf) -- the Oren-Nayar
// BSDF's tan(beta) and tan((alpha + beta)/2) with 0 <= beta <= alpha <= pi/2 (OrenNayarBsdf.cpp:95).  |x| >= 120: NaN (no call site).
PT_LIBM_FN float kernelTanf(float x, float y, int iy)
{
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f,
                T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f,
                T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f, T12 = 2.5907305826e-05f;
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {                          // |x| < 2^-13
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return 1.0f/__builtin_fabsf(x);
            else if (iy == 1) return x;
            else return -1.0f/x;
        }
    }
    if (ix >= 0x3f2ca140) {                         // |x| >= 0.6744
        if (hx < 0) { x = -x; y = -y; }
        const float z0 = pio4 - x;
        const float w0 = pio4lo - y;
        x = z0 + w0; y = 0.0f;
        if (__builtin_fabsf(x) < 0x1p-13f)
            return (float)(1 - ((hx >> 30) & 2))*(float)iy*(1.0f - 2.0f*(float)iy*x);
    }
    float z = x*x;
    float w = z*z;
    float r = T1 + w*(T3 + w*(T5 + w*(T7 + w*(T9 + w*T11))));
    float v = z*(T2 + w*(T4 + w*(T6 + w*(T8 + w*(T10 + w*T12)))));
    float s = z*x;
    r = y + z*(s*(r + v) + y);
    r += T0*s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2))*(v - 2.0f*(x - (w*w/(w + v) - r)));
    }
    if (iy == 1)
        return w;
    // -1/(x + r), accurately
    z = u2f(f2u(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f/w;
    const float t = u2f(f2u(a) & 0xfffff000u);
    s = 1.0f + t*z;
    return t + a*(s + t*v);
}
PT_LIBM_FN float tanfCore(float x)
{
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda)                           // |x| <= pi/4
        return kernelTanf(x, 0.0f, 1);
    if (!sincosInRange(x))                          // |x| >= 120 (and inf / NaN): glibc's large-argument reduction is not restated
        return u2f(0x7fc00000u);
    // the quadrant reduction of s_sincosf.h (reduce_fast) in double, WITHOUT fusing (tanf has no FMA variant); head and tail as floats
    const double hpiInv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double xd = (double)x;
    const int n = ((int32_t)(xd*hpiInv) + 0x800000) >> 24;
    const double y = xd - (double)n*hpi;
    const float head = (float)y;
    const float tail = (float)(y - (double)head);
    return kernelTanf(head, tail, 1 - ((n & 1) << 1));
}

// ---- double precision (round 6): exp, log, erf as glibc 2.35 computes them on an x86-64 host with FMA3 -----------------------------------
// media/AtmosphericMedium.cpp:113-122 inverts its optical depth in double: std::erf, std::exp, and -- inside Erf::erfInv -- std::log and std::sqrt.
// exp and log are the "optimized routines" algorithms (e_exp.c, e_log.c; N = 128 tables) in their __ieee754_exp_fma / __ieee754_log_fma builds:
// which operations those fuse was read off the disassembly of libm-2.35.a's e_exp-fma.o / e_log-fma.o (every a*b + c of the source is one fma;
// exp's subnormal branch keeps scale*tmp as a product because it is used twice).  erf is fdlibm's s_erf.c in glibc's regrouped evaluation
// (built without FMA: no variant of it exists), calling that exp.  Tables and coefficients: pt_libm_dtables.h, read out of the image's libm by
// tools/gen_libm_double_tables.py.  Matched against the host libm on 10^9 arguments per function (oracle/libm_host.cpp: libm_host_sweepd;
// tests/test_host.py), on the device by tests/test_gpu_libm.py.  sqrt is the hardware's / the compiler's correctly rounded one on both sides.
#include "pt_libm_dtables.h"

PT_LIBM_FN double expDSpecial(double tmp, uint64_t sbits, uint64_t ki)   // e_exp.c: specialcase (the result over- or underflows the scale's exponent)
{
    if ((ki & 0x80000000ull) == 0) {                                     // k > 0
        sbits -= 1009ull << 52;
        const double scale = u2d(sbits);
        return 0x1p1009*__builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;                                              // k < 0: care in the subnormal range
    const double scale = u2d(sbits);
    const double st = scale*tmp;
    double y = scale + st;
    if (y < 1.0) {
        double lo = (scale - y) + st;
        const double hi = 1.0 + y;
        lo = ((1.0 - hi) + y) + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022*y;
}
PT_LIBM_FN double expD(double x)
{
    const uint64_t ix = d2u(x);
    uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {                                      // |x| < 2^-54 or >= 512, inf, NaN
        if (abstop - 0x3c9u >= 0x80000000u)
            return 1.0 + x;
        if (abstop >= 0x409u) {                                          // |x| >= 1024
            if (ix == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (ix >> 63) ? 0.0 : u2d(0x7ff0000000000000ull);        // __math_uflow / __math_oflow
        }
        abstop = 0;                                                      // large |x|: through expDSpecial
    }
    const double Shift = g_expD[1];
    double kd = __builtin_fma(g_expD[0], x, Shift);
    const uint64_t ki = d2u(kd);
    kd -= Shift;
    double r = __builtin_fma(kd, g_expD[2], x);
    r = __builtin_fma(kd, g_expD[3], r);
    const uint64_t idx = 2u*(ki & 127u);
    const uint64_t top = ki << 45;
    const double tail = u2d(g_expDTab[idx]);
    const uint64_t sbits = g_expDTab[idx + 1] + top;
    const double r2 = r*r;
    const double p23 = __builtin_fma(r, g_expD[5], g_expD[4]);
    const double p45 = __builtin_fma(r, g_expD[7], g_expD[6]);
    double tmp = __builtin_fma(p23, r2, tail + r);
    tmp = __builtin_fma(r2*r2, p45, tmp);
    if (abstop == 0)
        return expDSpecial(tmp, sbits, ki);
    const double scale = u2d(sbits);
    return __builtin_fma(scale, tmp, scale);
}
PT_LIBM_FN double logD(double x)
{
    uint64_t ix = d2u(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    const double *A = g_logD + 2, *B = g_logD + 7;
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {               // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull)
            return 0.0;
        const double r = x - 1.0;
        const double r2 = r*r, r3 = r*r2;
        double a = __builtin_fma(r, B[2], B[1]), b = __builtin_fma(r, B[5], B[4]), c = __builtin_fma(r, B[8], B[7]);
        a = __builtin_fma(r2, B[3], a);
        b = __builtin_fma(r2, B[6], b);
        c = __builtin_fma(r2, B[9], c);
        c = __builtin_fma(r3, B[10], c);
        double P = __builtin_fma(c, r3, b);
        P = __builtin_fma(P, r3, a);
        const double w1 = __builtin_fma(r, 0x1p27, r);
        const double rhi = __builtin_fma(-0x1p27, r, w1);
        const double rlo = r - rhi;
        const double rhi2 = rhi*rhi;
        const double hi = __builtin_fma(rhi2, B[0], r);
        double lo = __builtin_fma(rhi2, B[0], r - hi);
        lo = __builtin_fma(B[0]*rlo, rhi + r, lo);
        const double y = __builtin_fma(P, r3, lo);
        return hi + y;
    }
    if (top - 0x10u >= 0x7ff0u - 0x10u) {                                // zero, subnormal, negative, inf, NaN
        if (ix*2 == 0)
            return -u2d(0x7ff0000000000000ull);
        if (ix == 0x7ff0000000000000ull)
            return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u)
            return u2d(0x7ff8000000000000ull)*((x != x) ? 1.0 : -1.0);   // __math_invalid: a NaN (its sign is not looked at by any caller)
        ix = d2u(x*0x1p52);
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
    const int32_t k = (int32_t)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = g_logDTab[i][0], logc = g_logDTab[i][1];
    const double z = u2d(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, g_logD[0], logc);
    const double hi = w + r;
    const double lo = __builtin_fma(kd, g_logD[1], (w - hi) + r);
    const double r2 = r*r, r3 = r*r2;
    const double p12 = __builtin_fma(r, A[2], A[1]), p34 = __builtin_fma(r, A[4], A[3]);
    const double t = __builtin_fma(r2, A[0], lo);
    const double q = __builtin_fma(p34, r2, p12);
    return __builtin_fma(r3, q, t) + hi;
}
PT_LIBM_FN double erfD(double x)                                         // s_erf.c: __erf
{
    const int32_t hx = (int32_t)(d2u(x) >> 32);
    const int32_t ix = hx & 0x7fffffff;
    const double *pp = g_erfD_pp, *qq = g_erfD_qq, *pa = g_erfD_pa, *qa = g_erfD_qa, *ra = g_erfD_ra, *sa = g_erfD_sa, *rb = g_erfD_rb, *sb = g_erfD_sb;
    if (ix >= 0x7ff00000) {                                              // erf(nan) = nan, erf(+-inf) = +-1
        const int32_t i = (int32_t)((uint32_t)hx >> 31) << 1;
        return (double)(1 - i) + 1.0/x;
    }
    if (ix < 0x3feb0000) {                                               // |x| < 0.84375
        if (ix < 0x3e300000) {                                           // |x| < 2^-28
            if (ix < 0x00800000)
                return 0.0625*(16.0*x + (16.0*g_erfD_efx)*x);
            return x + g_erfD_efx*x;
        }
        const double z = x*x;
        const double r1 = pp[0] + z*pp[1], z2 = z*z;
        const double r2 = pp[2] + z*pp[3], z4 = z2*z2;
        const double s1 = 1.0 + z*qq[1];
        const double s2 = qq[2] + z*qq[3];
        const double s3 = qq[4] + z*qq[5];
        const double r = r1 + z2*r2 + z4*pp[4];
        const double s = s1 + z2*s2 + z4*s3;
        const double y = r/s;
        return x + x*y;
    }
    if (ix < 0x3ff40000) {                                               // 0.84375 <= |x| < 1.25
        const double s = __builtin_fabs(x) - 1.0;
        const double P1 = pa[0] + s*pa[1], s2 = s*s;
        const double Q1 = 1.0 + s*qa[1], s4 = s2*s2;
        const double P2 = pa[2] + s*pa[3], s6 = s4*s2;
        const double Q2 = qa[2] + s*qa[3];
        const double P3 = pa[4] + s*pa[5];
        const double Q3 = qa[4] + s*qa[5];
        const double P4 = pa[6];
        const double Q4 = qa[6];
        const double P = P1 + s2*P2 + s4*P3 + s6*P4;
        const double Q = Q1 + s2*Q2 + s4*Q3 + s6*Q4;
        return hx >= 0 ? g_erfD_erx + P/Q : -g_erfD_erx - P/Q;
    }
    if (ix >= 0x40180000)                                                // |x| >= 6
        return hx >= 0 ? 1.0 - 1e-300 : 1e-300 - 1.0;
    const double ax = __builtin_fabs(x);
    const double s = 1.0/(ax*ax);
    double R, S;
    if (ix < 0x4006DB6E) {                                               // |x| < 1/0.35
        const double R1 = ra[0] + s*ra[1], s2 = s*s;
        const double S1 = 1.0 + s*sa[1], s4 = s2*s2;
        const double R2 = ra[2] + s*ra[3], s6 = s4*s2;
        const double S2 = sa[2] + s*sa[3], s8 = s4*s4;
        const double R3 = ra[4] + s*ra[5];
        const double S3 = sa[4] + s*sa[5];
        const double R4 = ra[6] + s*ra[7];
        const double S4 = sa[6] + s*sa[7];
        R = R1 + s2*R2 + s4*R3 + s6*R4;
        S = S1 + s2*S2 + s4*S3 + s6*S4 + s8*sa[8];
    } else {
        const double R1 = rb[0] + s*rb[1], s2 = s*s;
        const double S1 = 1.0 + s*sb[1], s4 = s2*s2;
        const double R2 = rb[2] + s*rb[3], s6 = s4*s2;
        const double S2 = sb[2] + s*sb[3];
        const double R3 = rb[4] + s*rb[5];
        const double S3 = sb[4] + s*sb[5];
        const double S4 = sb[6] + s*sb[7];
        R = R1 + s2*R2 + s4*R3 + s6*rb[6];
        S = S1 + s2*S2 + s4*S3 + s6*S4;
    }
    const double z = u2d(d2u(ax) & 0xffffffff00000000ull);
    const double r = expD(-z*z - 0.5625)*expD((z - ax)*(z + ax) + R/S);
    return hx >= 0 ? 1.0 - r/ax : r/ax - 1.0;
}

}  // namespace ptlibm

#endif
