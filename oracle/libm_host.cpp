// TEST INFRASTRUCTURE: tungsten_amd/csrc/hip/pt_libm.h (the device's restatements of glibc's sinf / cosf / logf / expf) compiled for
// the host, so that tests/test_host.py can hold the very text the kernels compile against the image's libm without a GPU.
// Built by the Makefile into oracle/libm_host.so (-mfma: the fused operations are single instructions, as on the device).
#include <stddef.h>
#define PT_LIBM_FN static inline
#define PT_LIBM_TABLE static const
#include "../tungsten_amd/csrc/hip/pt_libm.h"

// fn: 0 sinf, 1 cosf, 2 logf, 3 expf, 4 / 5 the sine / cosine of sincosfCore, 7 atanf, 8 cbrtf (6 is acosf: pt_math.h, a HIP header -- the
// device test covers it).  Arguments outside a function's range give NaN.
extern "C" void libm_host_eval(int fn, const float *x, float *y, size_t n)
{
    const float nan = __builtin_nanf("");
    for (size_t i = 0; i < n; ++i) {
        float s, c;
        switch (fn) {
        case 0: y[i] = ptlibm::sincosInRange(x[i]) ? ptlibm::sinfCore(x[i]) : nan; break;
        case 1: y[i] = ptlibm::sincosInRange(x[i]) ? ptlibm::cosfCore(x[i]) : nan; break;
        case 2: y[i] = ptlibm::logfAll(x[i]); break;
        case 3: y[i] = ptlibm::expfAll(x[i]); break;
        case 7: y[i] = ptlibm::atanfCore(x[i]); break;
        case 8: y[i] = ptlibm::cbrtfCore(x[i]); break;
        case 12: y[i] = ptlibm::tanfCore(x[i]); break;
        case 4: case 5:
            if (ptlibm::sincosInRange(x[i])) { ptlibm::sincosfCore(x[i], s, c); y[i] = fn == 4 ? s : c; } else y[i] = nan;
            break;
        default: y[i] = nan;
        }
    }
}

#include <math.h>
#include <string.h>
// the host libm itself over an array (fn as above; 6 = acosf; 9 = atan2f, 10 = powf over interleaved operand pairs): what the device's
// output is compared with
extern "C" void libm_host_ref(int fn, const float *x, float *y, size_t n)
{
    if (fn == 9 || fn == 10) {                       // the two-argument functions of tghip_debug_libm: operands interleaved, n results
        for (size_t i = 0; i < n; ++i)
            y[i] = fn == 9 ? atan2f(x[2*i], x[2*i + 1]) : powf(x[2*i], x[2*i + 1]);
        return;
    }
    for (size_t i = 0; i < n; ++i)
        y[i] = fn == 0 || fn == 4 ? sinf(x[i]) : fn == 1 || fn == 5 ? cosf(x[i]) : fn == 2 ? logf(x[i]) : fn == 3 ? expf(x[i]) : fn == 7 ? atanf(x[i])
             : fn == 8 ? cbrtf(x[i]) : fn == 12 ? tanf(x[i]) : acosf(x[i]);
}

// every stride-th float in [lo, hi] (as bit patterns, sign bit as given) against the host libm: returns the number of mismatches
extern "C" unsigned long long libm_host_sweep(int fn, unsigned int lo, unsigned int hi, unsigned int stride)
{
    unsigned long long bad = 0;
    const long long count = ((long long)hi - (long long)lo)/stride + 1;
#pragma omp parallel for reduction(+:bad)
    for (long long j = 0; j < count; ++j) {
        const long long b = (long long)lo + j*stride;
        unsigned int u = (unsigned int)b;
        float x, got, want, t;
        memcpy(&x, &u, 4);
        switch (fn) {
        case 0: if (!ptlibm::sincosInRange(x)) continue; got = ptlibm::sinfCore(x); want = sinf(x); break;
        case 1: if (!ptlibm::sincosInRange(x)) continue; got = ptlibm::cosfCore(x); want = cosf(x); break;
        case 2: got = ptlibm::logfAll(x); want = logf(x); if (got != got && want != want) continue; break;
        case 3: got = ptlibm::expfAll(x); want = expf(x); if (got != got && want != want) continue; break;
        case 7: got = ptlibm::atanfCore(x); want = atanf(x); if (got != got && want != want) continue; break;
        case 8: got = ptlibm::cbrtfCore(x); want = cbrtf(x); if (got != got && want != want) continue; break;
        case 12: { if ((u & 0x7fffffffu) > 0x3f490fdau && !ptlibm::sincosInRange(x)) continue; got = ptlibm::tanfCore(x); want = tanf(x); if (got != got && want != want) continue; break; }
        case 4: if (!ptlibm::sincosInRange(x)) continue; ptlibm::sincosfCore(x, got, t); want = sinf(x); break;
        default: if (!ptlibm::sincosInRange(x)) continue; ptlibm::sincosfCore(x, t, got); want = cosf(x); break;
        }
        if (memcmp(&got, &want, 4) != 0) bad++;
    }
    return bad;
}

// two-argument functions on n pseudo-random pairs (xorshift, 64 streams from `seed`): fn 0 atan2f -- a quarter of the pairs arbitrary bit
// patterns, the rest direction components in [-1, 1], also scaled to 1e-3 and 1e-4 --, fn 1 powf -- positive normal bases (a quarter of them
// in [1, 9): 1 + tau / p of the Davis transmittances), exponents in [-60, 60] and [-4, 4]; pairs glibc sends to its overflow / underflow
// paths are skipped.  Returns the number of mismatches.
static inline unsigned long long xorshift(unsigned long long *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
extern "C" unsigned long long libm_host_sweep2(int fn, unsigned long long n, unsigned long long seed, unsigned long long *tested)
{
    unsigned long long bad = 0, done = 0;
#pragma omp parallel for reduction(+:bad, done)
    for (int t = 0; t < 64; ++t) {
        unsigned long long s = 0x9E3779B97F4A7C15ull*(seed*64 + t + 1);
        for (unsigned long long i = 0; i < n/64; ++i) {
            const unsigned long long r = xorshift(&s);
            float x, y, got, want;
            unsigned int lo = (unsigned int)r, hi = (unsigned int)(r >> 32);
            if (fn == 0) {
                const int mode = (int)(i & 3);
                if (mode == 0) {
                    memcpy(&x, &lo, 4); memcpy(&y, &hi, 4);
                    // every 16th of these: one operand replaced by a value e_atan2f.c tests for (x == 1, zeros, infinities, exponents > 60 apart)
                    const int special = (int)((i >> 2) & 63);
                    if (special == 0) x = 1.0f; else if (special == 1) x = 0.0f; else if (special == 2) x = -0.0f; else if (special == 3) y = 0.0f;
                    else if (special == 4) y = -0.0f; else if (special == 5) x = __builtin_huge_valf(); else if (special == 6) x = -__builtin_huge_valf();
                    else if (special == 7) y = __builtin_huge_valf(); else if (special == 8) y = -__builtin_huge_valf();
                    else if (special == 9) { x = 1.0f; y = (float)((r & 0xffffff)/16777216.0)*4.0f - 2.0f; }
                    else if (special == 10) { x = 1e-25f*(float)((r & 0xffff) + 1); y = 3e12f; } else if (special == 11) { x = -3e20f; y = 1e-19f*(float)((r & 0xffff) + 1); }
                }
                else {
                    x = (float)((r & 0xffffff)/16777216.0)*2.0f - 1.0f;
                    y = (float)(((r >> 24) & 0xffffff)/16777216.0)*2.0f - 1.0f;
                    if (mode == 2) y *= 1e-3f;
                    if (mode == 3) x *= 1e-4f;
                }
                got = ptlibm::atan2fCore(y, x); want = atan2f(y, x);
                if (got != got && want != want) continue;
            } else {
                unsigned int bx = 0x00800000u + lo % 0x7f000000u;
                memcpy(&x, &bx, 4);
                if ((i & 3) == 2) x = 1.0f + (float)((r >> 8) & 0xffffff)/16777216.0f*8.0f;
                y = ((float)(hi & 0xffffff)/16777216.0f*2.0f - 1.0f)*((i & 1) ? 4.0f : 60.0f);
                if (!ptlibm::powInRange(x, y) || !ptlibm::powfCore(x, y, got)) continue;
                want = powf(x, y);
            }
            done++;
            if (memcmp(&got, &want, 4) != 0) bad++;
        }
    }
    if (tested) *tested = done;
    return bad;
}

// the host's RCPPS instruction over an array: what oracle/oracle.c: intel_rcpps restates (equal on Intel CPUs only; tests/test_host.py checks the vendor)
#include <xmmintrin.h>
extern "C" void libm_host_rcpps_hw(const float *x, float *y, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        y[i] = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(x[i])));
}

// ---- double precision (round 6): pt_libm.h's expD / logD / erfD against the host libm's exp / log / erf ------------------------------------
// fn: 0 exp, 1 log, 2 erf; libm_host_refd also 3 = sqrt (what the device's sqrt is compared with)
extern "C" void libm_host_evald(int fn, const double *x, double *y, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        y[i] = fn == 0 ? ptlibm::expD(x[i]) : fn == 1 ? ptlibm::logD(x[i]) : ptlibm::erfD(x[i]);
}
extern "C" void libm_host_refd(int fn, const double *x, double *y, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        y[i] = fn == 0 ? exp(x[i]) : fn == 1 ? log(x[i]) : fn == 2 ? erf(x[i]) : sqrt(x[i]);
}
// n pseudo-random arguments (xorshift, 64 streams from `seed`), a quarter each: arbitrary bit patterns; the range the function is interesting on
// (exp: [-750, 715]; log: (0, 2) and near one; erf: [-7, 7]); small magnitudes; the call sites' ranges (exp of -(s t)^2 - 0.5625 and of small
// negative numbers, log of q in (0, 1/2], erf of s t0 in [-4, 4]).  Returns the number of mismatches (NaN against NaN is a match).
extern "C" unsigned long long libm_host_sweepd(int fn, unsigned long long n, unsigned long long seed, unsigned long long *tested)
{
    unsigned long long bad = 0, done = 0;
#pragma omp parallel for reduction(+:bad, done)
    for (int t = 0; t < 64; ++t) {
        unsigned long long s = seed*0x9E3779B97F4A7C15ull + (unsigned long long)(t + 1)*0xD1B54A32D192ED03ull;
        for (int k = 0; k < 8; ++k) xorshift(&s);
        for (unsigned long long i = 0; i < n/64; ++i) {
            const unsigned long long r = xorshift(&s), r2 = xorshift(&s);
            const double u = (double)(r2 >> 11)*0x1p-53;           // [0, 1)
            double x;
            switch (i & 3) {
            case 0: memcpy(&x, &r, 8); break;
            case 1: x = fn == 0 ? u*1465.0 - 750.0 : fn == 1 ? ((r & 1) ? u*2.0 : 0.9 + u*0.2) : u*14.0 - 7.0; break;
            case 2: x = (u - 0.5)*((r & 1) ? 1e-3 : 1e-12); if (fn == 1) x = __builtin_fabs(x); break;
            default: x = fn == 0 ? ((r & 1) ? -u*40.0 : -u) : fn == 1 ? ((r & 1) ? u*0.5 : u*u*u*u*0.5) : u*8.0 - 4.0; break;
            }
            const double got = fn == 0 ? ptlibm::expD(x) : fn == 1 ? ptlibm::logD(x) : ptlibm::erfD(x);
            const double want = fn == 0 ? exp(x) : fn == 1 ? log(x) : erf(x);
            done++;
            if (got != got && want != want) continue;
            if (memcmp(&got, &want, 8) != 0) bad++;
        }
    }
    if (tested) *tested = done;
    return bad;
}
