/* Every float32 bit pattern through the restatement of Intel's RCPPS (oracle/oracle.c: intel_rcpps -- the text below is that function)
 * against the instruction itself.  Meaningful on an Intel CPU only (AMD's RCPPS is another function): prints the number of mismatches.
 *     gcc -O2 -fopenmp -ffp-contract=off tools/rcpps_sweep.c -o /tmp/rcpps_sweep && /tmp/rcpps_sweep
 * On the Xeon the goldens of this repository were rendered on: "mismatches: 0 of 4294967296" (profiles/r4_rcpps_sweep.txt). */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <xmmintrin.h>

static inline float intel_rcpps(float x)
{
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u, e = (u >> 23) & 0xffu, i = (u >> 12) & 0x7ffu;
    uint32_t bits;
    if (e == 0u) bits = sign | 0x7f800000u;
    else if (e == 255u) bits = (u & 0x7fffffu) ? (u | 0x00400000u) : sign;
    else if (e >= 253u) bits = sign;
    else {
        const uint32_t d = 4097u + 2u*i;
        int32_t q = (int32_t)(33554432.0f/(float)d);
        int32_t r = (int32_t)(33554432u - (uint32_t)q*d);
        if (r < 0) { q -= 1; r += (int32_t)d; }
        if (r >= (int32_t)d) { q += 1; r -= (int32_t)d; }
        if (2*r > (int32_t)d) q += 1;
        bits = sign | ((253u - e) << 23) | ((uint32_t)(q - 4096) << 11);
    }
    float f; memcpy(&f, &bits, 4);
    return f;
}

int main(void)
{
    unsigned long long bad = 0;
#pragma omp parallel for reduction(+:bad)
    for (long long b = 0; b < (1ll << 32); ++b) {
        uint32_t u = (uint32_t)b, hu, ru;
        float x, h, r;
        memcpy(&x, &u, 4);
        h = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(x)));
        r = intel_rcpps(x);
        memcpy(&hu, &h, 4); memcpy(&ru, &r, 4);
        if (hu != ru) bad++;
    }
    printf("mismatches: %llu of 4294967296\n", bad);
    return bad != 0;
}
