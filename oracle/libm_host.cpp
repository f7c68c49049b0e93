// TEST INFRASTRUCTURE: tungsten_amd/csrc/hip/pt_libm.h (the device's restatements of glibc's sinf / cosf / logf / expf) compiled for
// the host, so that tests/test_host.py can hold the very text the kernels compile against the image's libm without a GPU.
// Built by the Makefile into oracle/libm_host.so (-mfma: the fused operations are single instructions, as on the device).
#include <stddef.h>
#define PT_LIBM_FN static inline
#define PT_LIBM_TABLE static const
#include "../tungsten_amd/csrc/hip/pt_libm.h"

// fn: 0 sinf, 1 cosf, 2 logf, 3 expf, 4 / 5 the sine / cosine of sincosfCore.  Arguments outside a function's range give NaN.
extern "C" void libm_host_eval(int fn, const float *x, float *y, size_t n)
{
    const float nan = __builtin_nanf("");
    for (size_t i = 0; i < n; ++i) {
        float s, c;
        switch (fn) {
        case 0: y[i] = ptlibm::sincosInRange(x[i]) ? ptlibm::sinfCore(x[i]) : nan; break;
        case 1: y[i] = ptlibm::sincosInRange(x[i]) ? ptlibm::cosfCore(x[i]) : nan; break;
        case 2: y[i] = ptlibm::logInRange(x[i]) ? ptlibm::logfCore(x[i]) : nan; break;
        case 3: y[i] = ptlibm::expInRange(x[i]) ? ptlibm::expfCore(x[i]) : nan; break;
        case 4: case 5:
            if (ptlibm::sincosInRange(x[i])) { ptlibm::sincosfCore(x[i], s, c); y[i] = fn == 4 ? s : c; } else y[i] = nan;
            break;
        default: y[i] = nan;
        }
    }
}

#include <math.h>
#include <string.h>
// the host libm itself over an array (fn as above; 6 = acosf): what the device's output is compared with
extern "C" void libm_host_ref(int fn, const float *x, float *y, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        y[i] = fn == 0 || fn == 4 ? sinf(x[i]) : fn == 1 || fn == 5 ? cosf(x[i]) : fn == 2 ? logf(x[i]) : fn == 3 ? expf(x[i]) : acosf(x[i]);
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
        case 2: if (!ptlibm::logInRange(x)) continue; got = ptlibm::logfCore(x); want = logf(x); break;
        case 3: if (!ptlibm::expInRange(x)) continue; got = ptlibm::expfCore(x); want = expf(x); break;
        case 4: if (!ptlibm::sincosInRange(x)) continue; ptlibm::sincosfCore(x, got, t); want = sinf(x); break;
        default: if (!ptlibm::sincosInRange(x)) continue; ptlibm::sincosfCore(x, t, got); want = cosf(x); break;
        }
        if (memcmp(&got, &want, 4) != 0) bad++;
    }
    return bad;
}
