/*
 * oracle/oracle.c -- TEST INFRASTRUCTURE.  NOT PRODUCT CODE.
 *
 * A scalar CPU restatement of the reference's forward path tracer (tunabrain/tungsten,
 * src/core/integrators/path_tracer + TraceBase + the BSDF/primitive/texture/sampling code it
 * calls), written in plain C99 directly from the reference's algorithm, each function citing
 * the reference file:line it follows.  It consumes the same flattened scene description the
 * HIP path uploads (include/tungsten_hip.h), so checker and product see identical inputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (tungsten_amd/, csrc/) never links, imports or calls it.
 *
 * Pinning: tests/golden/ holds outputs of the *reference itself* (built from /root/reference by
 * oracle/Makefile.ref and driven by oracle/ref_harness.cpp, which injects the same
 * counter-based random stream into the reference's PathTracer::traceSample); tests/test_oracle_*.py
 * check this file against them.  See DESIGN.md "Oracle".  State at the end of round 4: the per-sample radiance is the reference's bit for bit
 * (float32 ==, three channels) in every one of the 67 golden cases, 601 344 samples, and in the 13 lifted twins; the per-pass records of the
 * reference's own integrator loop and its five output buffers likewise (tests/test_oracle_golden.py, test_adaptive_cpu.py, test_outputs_cpu.py).
 * Two paths in here are test-side prototypes and off in every test of the device: oracle_set_top_items, flat_shortcut_decides_v2.
 *
 * Numerics: float everywhere, the reference's constants (PI = 3.1415926536f, math/Angle.hpp:8),
 * same operation order where it matters; compiled with -ffp-contract=off.
 */
#include "../include/tungsten_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define O_PI          3.1415926536f
#define O_TWO_PI      (O_PI*2.0f)
#define O_FOUR_PI     (O_PI*4.0f)
#define O_INV_PI      (1.0f/O_PI)
#define O_INV_TWO_PI  (0.5f*O_INV_PI)
#define O_INV_FOUR_PI (0.25f*O_INV_PI)

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vs(float s) { return V(s, s, s); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x*b.x, a.y*b.y, a.z*b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V(a.x/b.x, a.y/b.y, a.z/b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x*s, a.y*s, a.z*s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x/s, a.y/s, a.z/s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
static inline v3 vcross(v3 a, v3 b) { return V(a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); }
static inline float vlensq(v3 a) { return a.x*a.x + a.y*a.y + a.z*a.z; }
static inline float vlen(v3 a) { return sqrtf(vlensq(a)); }
static inline v3 vnorm(v3 a) { float inv = 1.0f/vlen(a); return V(a.x*inv, a.y*inv, a.z*inv); }   /* Vec.hpp:168-175 */
static inline float vmax3(v3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
static inline float vavg(v3 a) { return (a.x + a.y + a.z)*(1.0f/3.0f); }                           /* Vec.hpp avg() */
static inline float vsum(v3 a) { return a.x + a.y + a.z; }
static inline int viszero(v3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }            /* Vec == scalar: all components (Vec.hpp:429-435) */
static inline v3 vexp(v3 a) { return V(expf(a.x), expf(a.y), expf(a.z)); }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline float sqr(float x) { return x*x; }
static inline float fclamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* ---------------------------------------------------------------------------------------------
 * Random numbers.  The reference draws from one sequential PCG stream per 16x16 tile
 * (PathTraceIntegrator.cpp:27-42, UniformPathSampler.hpp), which cannot be reproduced by a
 * parallel renderer.  Oracle and HIP path (and oracle/ref_harness.cpp, which injects it into the
 * reference) use the same PCG-XSH-RR 64/32 generator (UniformSampler.hpp:40-47) but keyed per
 * (seed, pixelIndex, sampleIndex): DESIGN.md "RNG".
 * ------------------------------------------------------------------------------------------- */
static inline uint32_t hash32(uint32_t x)   /* MathUtil.hpp:120-128 */
{
    x = ~x + (x << 15);
    x = x ^ (x >> 12);
    x = x + (x << 2);
    x = x ^ (x >> 4);
    x = x * 2057;
    x = x ^ (x >> 16);
    return x;
}

typedef struct {
    uint64_t state, inc;
    const float *replay;   /* when non-NULL, next1D() returns these numbers instead (unit tests) */
    int replay_pos, replay_n;
    uint32_t draws;
    /* SobolPathSampler state (sampling/SobolPathSampler.hpp:14-18); sobol == NULL: uniform sampler */
    const uint32_t *sobol;
    uint32_t scramble, index, dimension;
    float *log; int log_n, log_cap;   /* debugging: every number drawn (oracle_trace_sample_log) */
} Sampler;
static inline float sampler_log(Sampler *s, float v) { if (s->log && s->log_n < s->log_cap) s->log[s->log_n++] = v; return v; }

static void sampler_start(Sampler *s, uint32_t seed, uint32_t pixelIndex, uint32_t sampleIndex)
{
    uint32_t a = hash32(seed) ^ pixelIndex;
    uint32_t b = hash32(a) + sampleIndex;
    uint32_t hi = hash32(b), lo = hash32(b ^ 0x9E3779B9u);
    s->state = ((uint64_t)hi << 32) | lo;
    s->inc = ((uint64_t)pixelIndex << 1) | 1u;
    s->replay = NULL; s->replay_pos = s->replay_n = 0; s->draws = 0;
    s->sobol = NULL; s->scramble = s->index = s->dimension = 0;
    s->log = NULL; s->log_n = s->log_cap = 0;
}

/* SobolPathSampler::startPath (SobolPathSampler.hpp:47-52); the supplemental stream (booleans, dimensions >= 1024)
 * is the counter-based one above instead of the reference's sequential per-tile UniformSampler */
static void sampler_start_sobol(Sampler *s, const uint32_t *matrices, uint32_t tileSeed, uint32_t seed, uint32_t pixelIndex, uint32_t sampleIndex)
{
    sampler_start(s, seed, pixelIndex, sampleIndex);
    s->sobol = matrices;
    s->scramble = tileSeed ^ hash32(pixelIndex);
    s->index = sampleIndex;
    s->dimension = 0;
}

/* sobol::sample (thirdparty/sobol/sobol.h:39-53): XOR of the generator-matrix columns selected by the index bits */
static inline uint32_t sobol_sample(const uint32_t *matrices, uint64_t index, uint32_t dimension, uint32_t scramble)
{
    uint32_t result = scramble;
    for (uint32_t i = dimension*TGHIP_SOBOL_BITS; index; index >>= 1, ++i)
        if (index & 1)
            result ^= matrices[i];
    return result;
}

static inline uint32_t sampler_nextI(Sampler *s)   /* UniformSampler.hpp:40-47 */
{
    uint64_t oldState = s->state;
    s->state = oldState*6364136223846793005ULL + s->inc;
    uint32_t xorShifted = (uint32_t)(((oldState >> 18u) ^ oldState) >> 27u);
    uint32_t rot = (uint32_t)(oldState >> 59u);
    return (xorShifted >> rot) | (xorShifted << ((uint32_t)(-(int32_t)rot) & 31));
}

static inline float normalizedUint(uint32_t i)   /* BitManip.hpp:47-50 */
{
    union { uint32_t u; float f; } c;
    c.u = (i >> 9u) | 0x3F800000u;
    return c.f - 1.0f;
}

static inline float nextSupplemental(Sampler *s)
{
    s->draws++;
    if (s->replay) {
        float v = s->replay_pos < s->replay_n ? s->replay[s->replay_pos] : 0.5f;
        s->replay_pos++;
        return v;
    }
    return sampler_log(s, normalizedUint(sampler_nextI(s)));
}
static inline float next1D(Sampler *s)   /* UniformPathSampler.hpp:44-47 / SobolPathSampler.hpp:64-69 */
{
    if (s->sobol && s->dimension < TGHIP_SOBOL_DIMS) {
        uint32_t permuted = (s->index & ~0xFFu) | ((s->index + s->scramble) & 0xFFu);   /* permutedIndex(), :20-23 */
        s->draws++;
        return sampler_log(s, normalizedUint(sobol_sample(s->sobol, permuted, s->dimension++, s->scramble)));
    }
    return nextSupplemental(s);
}
/* UniformPathSampler.hpp:39-42 / SobolPathSampler.hpp:54-57 (always the supplemental stream) */
static inline int nextBoolean(Sampler *s, float pTrue) { return nextSupplemental(s) < pTrue; }

/* ---------------------------------------------------------------------------------------------
 * Sample warps (sampling/SampleWarp.hpp)
 * ------------------------------------------------------------------------------------------- */
static v3 cosineHemisphere(float xi0, float xi1)   /* SampleWarp.hpp:42-52 */
{
    float phi = xi0*O_TWO_PI;
    float r = sqrtf(xi1);
    return V(cosf(phi)*r, sinf(phi)*r, sqrtf(fmaxf(1.0f - xi1, 0.0f)));
}
static inline float cosineHemispherePdf(v3 p) { return fabsf(p.z)*O_INV_PI; }   /* :54-57 */
static v3 uniformHemisphere(float xi0, float xi1)  /* SampleWarp.hpp:25-30 */
{
    float phi = O_TWO_PI*xi0;
    float r = sqrtf(fmaxf(1.0f - xi1*xi1, 0.0f));
    return V(cosf(phi)*r, sinf(phi)*r, xi1);
}
static v3 uniformSphere(float xi0, float xi1)      /* SampleWarp.hpp:96-107 */
{
    float phi = xi0*O_TWO_PI;
    float z = xi1*2.0f - 1.0f;
    float r = sqrtf(fmaxf(1.0f - z*z, 0.0f));
    return V(cosf(phi)*r, sinf(phi)*r, z);
}
static inline float powerHeuristic(float pdf0, float pdf1) { return (pdf0*pdf0)/(pdf0*pdf0 + pdf1*pdf1); }   /* :189-192 */

/* TangentFrame(n) -- Duff et al. ONB (math/TangentFrame.hpp:22-31) */
typedef struct { v3 normal, tangent, bitangent; } Frame;
static Frame frame_from_normal(v3 n)
{
    Frame f;
    f.normal = n;
    float sign = copysignf(1.0f, n.z);
    const float a = -1.0f/(sign + n.z);
    const float b = n.x*n.y*a;
    f.tangent = V(1.0f + sign*n.x*n.x*a, sign*b, -sign*n.x);
    f.bitangent = V(b, sign + n.y*n.y*a, -n.y);
    return f;
}
static inline v3 toLocal(const Frame *f, v3 p) { return V(vdot(f->tangent, p), vdot(f->bitangent, p), vdot(f->normal, p)); }
static inline v3 toGlobal(const Frame *f, v3 p)
{
    return vadd(vadd(vscale(f->tangent, p.x), vscale(f->bitangent, p.y)), vscale(f->normal, p.z));
}
static inline v3 mat3_mul(const float *m, v3 p)      /* row-major 3x3 times vector (Mat4f::transformVector) */
{
    return V(m[0]*p.x + m[1]*p.y + m[2]*p.z, m[3]*p.x + m[4]*p.y + m[5]*p.z, m[6]*p.x + m[7]*p.y + m[8]*p.z);
}
static inline v3 mat3_tmul(const float *m, v3 p)     /* transpose(m) times vector (= _invRot*p) */
{
    return V(m[0]*p.x + m[3]*p.y + m[6]*p.z, m[1]*p.x + m[4]*p.y + m[7]*p.z, m[2]*p.x + m[5]*p.y + m[8]*p.z);
}

/* ---------------------------------------------------------------------------------------------
 * Textures (textures/ConstantTexture, CheckerTexture.cpp:64-69, BitmapTexture.cpp:298-352)
 * ------------------------------------------------------------------------------------------- */
static v3 bitmap_texel(const TgHipSceneDesc *s, const TgHipTexture *t, int x, int y)
{
    const float *tex = s->texels + t->texel_offset;
    if (t->flags & TGHIP_TEXF_RGB) {
        const float *p = tex + ((size_t)x + (size_t)y*t->w)*3;
        return V(p[0], p[1], p[2]);
    } else {
        return vs(tex[(size_t)x + (size_t)y*t->w]);
    }
}

static v3 texture_eval(const TgHipSceneDesc *s, int texIdx, float u0, float v0)
{
    const TgHipTexture *t = &s->textures[texIdx];
    if (t->type == TGHIP_TEX_CONSTANT)
        return ld3(t->value);
    if (t->type == TGHIP_TEX_CHECKER) {
        int ui = (int)(u0*(float)t->res_u), vi = (int)(v0*(float)t->res_v);
        int on = (ui ^ vi) & 1;
        return on ? ld3(t->on_color) : ld3(t->off_color);
    }
    /* bitmap */
    int w = t->w, h = t->h;
    float u = u0*w;
    float v = (1.0f - v0)*h;
    int linear = (t->flags & TGHIP_TEXF_LINEAR) && (t->flags & TGHIP_TEXF_VALID);
    if (linear) { u -= 0.5f; v -= 0.5f; }
    int iu0 = u < 0.0f ? -(int)(-u) - 1 : (int)u;
    int iv0 = v < 0.0f ? -(int)(-v) - 1 : (int)v;
    int iu1 = iu0 + 1, iv1 = iv0 + 1;
    u -= iu0; v -= iv0;
    if (!(t->flags & TGHIP_TEXF_CLAMP)) {
        iu0 = ((iu0 % w) + w) % w; iu1 = ((iu1 % w) + w) % w;
        iv0 = ((iv0 % h) + h) % h; iv1 = ((iv1 % h) + h) % h;
    } else {
        iu0 = iu0 < 0 ? 0 : (iu0 > w - 1 ? w - 1 : iu0); iu1 = iu1 < 0 ? 0 : (iu1 > w - 1 ? w - 1 : iu1);
        iv0 = iv0 < 0 ? 0 : (iv0 > h - 1 ? h - 1 : iv0); iv1 = iv1 < 0 ? 0 : (iv1 > h - 1 ? h - 1 : iv1);
    }
    if (!linear)
        return bitmap_texel(s, t, iu0, iv0);      /* sic: unfiltered lookups ignore _scale (BitmapTexture.cpp:327-332) */
    v3 x00 = bitmap_texel(s, t, iu0, iv0), x01 = bitmap_texel(s, t, iu1, iv0);
    v3 x10 = bitmap_texel(s, t, iu0, iv1), x11 = bitmap_texel(s, t, iu1, iv1);
    v3 top = vadd(vscale(x00, 1.0f - u), vscale(x01, u));
    v3 bot = vadd(vscale(x10, 1.0f - u), vscale(x11, u));
    v3 r = vadd(vscale(top, 1.0f - v), vscale(bot, v));
    return vscale(r, t->scale);
}

/* Distribution2D::warp / pdf (sampling/Distribution2D.hpp:68-83) over the flattened tables */
typedef struct { const float *mpdf, *mcdf, *pdf, *cdf; int w, h; } Dist2D;
static Dist2D dist_of(const TgHipSceneDesc *s, const TgHipTexture *t)
{
    Dist2D d;
    d.w = t->w; d.h = t->h;
    d.mpdf = s->dist + t->dist_offset;
    d.mcdf = d.mpdf + t->h;
    d.pdf = d.mcdf + t->h + 1;
    d.cdf = d.pdf + (size_t)t->w*t->h;
    return d;
}
static int upper_bound_idx(const float *a, int n, float x)   /* std::upper_bound: first element > x */
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
static void dist_warp(const Dist2D *d, float *u, float *v, int *row, int *column)
{
    *row = upper_bound_idx(d->mcdf, d->h + 1, *v) - 1;
    *v = fclamp((*v - d->mcdf[*row])/d->mpdf[*row], 0.0f, 1.0f);
    const float *rowStart = d->cdf + (size_t)(*row)*(d->w + 1);
    *column = upper_bound_idx(rowStart, d->w + 1, *u) - 1;
    int idxC = *row*(d->w + 1) + *column;
    int idxP = *row*d->w + *column;
    *u = fclamp((*u - d->cdf[idxC])/d->pdf[idxP], 0.0f, 1.0f);
}
static float dist_pdf(const Dist2D *d, int row, int column)
{
    row = row < 0 ? 0 : (row > d->h - 1 ? d->h - 1 : row);
    column = column < 0 ? 0 : (column > d->w - 1 ? d->w - 1 : column);
    return d->pdf[(size_t)row*d->w + column]*d->mpdf[row];
}
/* BitmapTexture::sample / pdf (BitmapTexture.cpp:433-455) */
static void bitmap_sample(const TgHipSceneDesc *s, const TgHipTexture *t, float xi0, float xi1, float *u, float *v)
{
    Dist2D d = dist_of(s, t);
    int row, column;
    float nu = xi0, nv = xi1;
    dist_warp(&d, &nu, &nv, &row, &column);
    *u = (nu + column)/t->w;
    *v = 1.0f - (nv + row)/t->h;
}
static float bitmap_pdf(const TgHipSceneDesc *s, const TgHipTexture *t, float u, float v)
{
    Dist2D d = dist_of(s, t);
    return dist_pdf(&d, (int)((1.0f - v)*t->h), (int)(u*t->w))*t->w*t->h;
}

/* ---------------------------------------------------------------------------------------------
 * Fresnel + microfacet helpers (bsdfs/Fresnel.hpp:75-138, bsdfs/Microfacet.hpp:27-130)
 * ------------------------------------------------------------------------------------------- */
static float dielectricReflectanceT(float eta, float cosThetaI, float *cosThetaT)
{
    if (cosThetaI < 0.0f) {
        eta = 1.0f/eta;
        cosThetaI = -cosThetaI;
    }
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f) {
        *cosThetaT = 0.0f;
        return 1.0f;
    }
    *cosThetaT = sqrtf(fmaxf(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta*cosThetaI - *cosThetaT)/(eta*cosThetaI + *cosThetaT);
    float Rp = (eta**cosThetaT - cosThetaI)/(eta**cosThetaT + cosThetaI);
    return (Rs*Rs + Rp*Rp)*0.5f;
}
static float dielectricReflectance(float eta, float cosThetaI) { float t; return dielectricReflectanceT(eta, cosThetaI, &t); }
/* Fresnel::thinFilmReflectance (Fresnel.hpp:15-28) */
static float thinFilmReflectance(float eta, float cosThetaI, float *cosThetaT)
{
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f) {
        *cosThetaT = 0.0f;
        return 1.0f;
    }
    *cosThetaT = sqrtf(fmaxf(1.0f - sinThetaTSq, 0.0f));
    float Rs = sqr((eta*cosThetaI - *cosThetaT)/(eta*cosThetaI + *cosThetaT));
    float Rp = sqr((eta*(*cosThetaT) - cosThetaI)/(eta*(*cosThetaT) + cosThetaI));
    return 1.0f - ((1.0f - Rs)/(1.0f + Rs) + (1.0f - Rp)/(1.0f + Rp))*0.5f;
}
/* Fresnel::thinFilmReflectanceInterference (Fresnel.hpp:39-67): `thickness` in nanometres */
static v3 thinFilmReflectanceInterference(float eta, float cosThetaI, float thickness, float *cosThetaT)
{
    const v3 invLambdas = V(1.0f/650.0f, 1.0f/510.0f, 1.0f/475.0f);
    float cosThetaISq = cosThetaI*cosThetaI;
    float sinThetaISq = 1.0f - cosThetaISq;
    float invEta = 1.0f/eta;
    float sinThetaTSq = eta*eta*sinThetaISq;
    if (sinThetaTSq > 1.0f) {
        *cosThetaT = 0.0f;
        return vs(1.0f);
    }
    *cosThetaT = sqrtf(1.0f - sinThetaTSq);
    float Ts = 4.0f*eta*cosThetaI*(*cosThetaT)/sqr(eta*cosThetaI + *cosThetaT);
    float Tp = 4.0f*eta*cosThetaI*(*cosThetaT)/sqr(eta*(*cosThetaT) + cosThetaI);
    float Rs = 1.0f - Ts;
    float Rp = 1.0f - Tp;
    v3 phi = vscale(invLambdas, thickness*(*cosThetaT)*O_FOUR_PI*invEta);
    v3 cosPhi = V(cosf(phi.x), cosf(phi.y), cosf(phi.z));
    float a = sqr(Rs) + 1.0f, b2 = 2.0f*Rs, c = sqr(Rp) + 1.0f, d2 = 2.0f*Rp, ts = sqr(Ts), tp = sqr(Tp);
    v3 tS = V(ts/(a - b2*cosPhi.x), ts/(a - b2*cosPhi.y), ts/(a - b2*cosPhi.z));
    v3 tP = V(tp/(c - d2*cosPhi.x), tp/(c - d2*cosPhi.y), tp/(c - d2*cosPhi.z));
    return V(1.0f - (tS.x + tP.x)*0.5f, 1.0f - (tS.y + tP.y)*0.5f, 1.0f - (tS.z + tP.z)*0.5f);
}

static float conductorReflectance1(float eta, float k, float cosThetaI)
{
    float cosThetaISq = cosThetaI*cosThetaI;
    float sinThetaISq = fmaxf(1.0f - cosThetaISq, 0.0f);
    float sinThetaIQu = sinThetaISq*sinThetaISq;
    float innerTerm = eta*eta - k*k - sinThetaISq;
    float aSqPlusBSq = sqrtf(fmaxf(innerTerm*innerTerm + 4.0f*eta*eta*k*k, 0.0f));
    float a = sqrtf(fmaxf((aSqPlusBSq + innerTerm)*0.5f, 0.0f));
    float Rs = ((aSqPlusBSq + cosThetaISq) - (2.0f*a*cosThetaI))/
               ((aSqPlusBSq + cosThetaISq) + (2.0f*a*cosThetaI));
    float Rp = ((cosThetaISq*aSqPlusBSq + sinThetaIQu) - (2.0f*a*cosThetaI*sinThetaISq))/
               ((cosThetaISq*aSqPlusBSq + sinThetaIQu) + (2.0f*a*cosThetaI*sinThetaISq));
    return 0.5f*(Rs + Rs*Rp);
}
static v3 conductorReflectance(const float *eta, const float *k, float cosThetaI)
{
    return V(conductorReflectance1(eta[0], k[0], cosThetaI), conductorReflectance1(eta[1], k[1], cosThetaI),
             conductorReflectance1(eta[2], k[2], cosThetaI));
}

static float mf_roughnessToAlpha(int dist, float roughness)
{
    const float MinAlpha = 1e-3f;
    roughness = fmaxf(roughness, MinAlpha);
    if (dist == TGHIP_DIST_PHONG)
        return 2.0f/(roughness*roughness) - 2.0f;
    return roughness;
}
static float mf_D(int dist, float alpha, v3 m)
{
    if (m.z <= 0.0f)
        return 0.0f;
    switch (dist) {
    case TGHIP_DIST_BECKMANN: {
        float alphaSq = alpha*alpha;
        float cosThetaSq = m.z*m.z;
        float tanThetaSq = fmaxf(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
        float cosThetaQu = cosThetaSq*cosThetaSq;
        return O_INV_PI*expf(-tanThetaSq/alphaSq)/(alphaSq*cosThetaQu);
    }
    case TGHIP_DIST_PHONG:
        return (alpha + 2.0f)*O_INV_TWO_PI*(float)pow((double)m.z, (double)alpha);
    case TGHIP_DIST_GGX: {
        float alphaSq = alpha*alpha;
        float cosThetaSq = m.z*m.z;
        float tanThetaSq = fmaxf(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
        float cosThetaQu = cosThetaSq*cosThetaSq;
        return alphaSq*O_INV_PI/(cosThetaQu*sqr(alphaSq + tanThetaSq));
    }
    }
    return 0.0f;
}
static float mf_G1(int dist, float alpha, v3 v, v3 m)
{
    if (vdot(v, m)*v.z <= 0.0f)
        return 0.0f;
    switch (dist) {
    case TGHIP_DIST_BECKMANN: {
        float cosThetaSq = v.z*v.z;
        float tanTheta = fabsf(sqrtf(fmaxf(1.0f - cosThetaSq, 0.0f))/v.z);
        float a = 1.0f/(alpha*tanTheta);
        if (a < 1.6f)
            return (3.535f*a + 2.181f*a*a)/(1.0f + 2.276f*a + 2.577f*a*a);
        return 1.0f;
    }
    case TGHIP_DIST_PHONG: {
        float cosThetaSq = v.z*v.z;
        float tanTheta = fabsf(sqrtf(fmaxf(1.0f - cosThetaSq, 0.0f))/v.z);
        float a = sqrtf(0.5f*alpha + 1.0f)/tanTheta;
        if (a < 1.6f)
            return (3.535f*a + 2.181f*a*a)/(1.0f + 2.276f*a + 2.577f*a*a);
        return 1.0f;
    }
    case TGHIP_DIST_GGX: {
        float alphaSq = alpha*alpha;
        float cosThetaSq = v.z*v.z;
        float tanThetaSq = fmaxf(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
        return 2.0f/(1.0f + sqrtf(1.0f + alphaSq*tanThetaSq));
    }
    }
    return 0.0f;
}
static float mf_G(int dist, float alpha, v3 i, v3 o, v3 m) { return mf_G1(dist, alpha, i, m)*mf_G1(dist, alpha, o, m); }
static float mf_pdf(int dist, float alpha, v3 m) { return mf_D(dist, alpha, m)*m.z; }
static v3 mf_sample(int dist, float alpha, float xi0, float xi1)
{
    float phi = xi1*O_TWO_PI;
    float cosTheta = 0.0f;
    switch (dist) {
    case TGHIP_DIST_BECKMANN: {
        float tanThetaSq = -alpha*alpha*logf(1.0f - xi0);
        cosTheta = 1.0f/sqrtf(1.0f + tanThetaSq);
        break;
    }
    case TGHIP_DIST_PHONG:
        cosTheta = (float)pow((double)xi0, 1.0/((double)alpha + 2.0));
        break;
    case TGHIP_DIST_GGX: {
        float tanThetaSq = alpha*alpha*xi0/(1.0f - xi0);
        cosTheta = 1.0f/sqrtf(1.0f + tanThetaSq);
        break;
    }
    }
    float r = sqrtf(fmaxf(1.0f - cosTheta*cosTheta, 0.0f));
    return V(cosf(phi)*r, sinf(phi)*r, cosTheta);
}

/* ---------------------------------------------------------------------------------------------
 * BSDFs.  One event struct (samplerecords/SurfaceScatterEvent.hpp:14-44) and three entry
 * points per BSDF, dispatched on the tagged union; nested BSDFs recurse.
 * ------------------------------------------------------------------------------------------- */
#define LOBE_ALL            (TGHIP_LOBE_GLOSSY_R | TGHIP_LOBE_GLOSSY_T | TGHIP_LOBE_DIFFUSE_R | TGHIP_LOBE_DIFFUSE_T | \
                             TGHIP_LOBE_SPECULAR_R | TGHIP_LOBE_SPECULAR_T | TGHIP_LOBE_ANISOTROPIC)   /* BsdfLobes.hpp:28-31 */
#define LOBE_SPECULAR       (TGHIP_LOBE_SPECULAR_R | TGHIP_LOBE_SPECULAR_T)
#define LOBE_TRANSMISSIVE   (TGHIP_LOBE_GLOSSY_T | TGHIP_LOBE_DIFFUSE_T | TGHIP_LOBE_SPECULAR_T)
#define LOBE_ALL_BUT_SPECULAR (~(uint32_t)(LOBE_SPECULAR | TGHIP_LOBE_FORWARD))                     /* BsdfLobes.hpp:32 */

typedef struct {
    v3 wi, wo, weight;
    float pdf;
    uint32_t requested, sampled;
    float u, v;              /* info->uv for texture lookups */
    Sampler *sampler;
} Event;

static const float DiracAcceptanceThreshold = 1e-3f;   /* Bsdf.hpp:27 */
static int checkReflectionConstraint(v3 wi, v3 wo)     /* Bsdf.hpp:45-48 */
{
    return fabsf(wi.z*wo.z - wi.x*wo.x - wi.y*wo.y - 1.0f) < DiracAcceptanceThreshold;
}
static int checkRefractionConstraint(v3 wi, v3 wo, float eta, float cosThetaT)   /* Bsdf.hpp:50-54 */
{
    float dotP = -wi.x*wo.x*eta - wi.y*wo.y*eta - copysignf(cosThetaT, wi.z)*wo.z;
    return fabsf(dotP - 1.0f) < DiracAcceptanceThreshold;
}
static inline float sgnE(float v) { return v < 0.0f ? -1.0f : 1.0f; }   /* RoughDielectricBsdf.cpp:13-16 */

static v3 bsdf_albedo(const TgHipSceneDesc *s, const TgHipBsdf *b, const Event *e) { return texture_eval(s, b->albedo, e->u, e->v); }
static float bsdf_roughness(const TgHipSceneDesc *s, const TgHipBsdf *b, const Event *e) { return texture_eval(s, b->roughness, e->u, e->v).x; }

static v3 bsdf_eval(const TgHipSceneDesc *s, int bi, const Event *e);
static int bsdf_sample(const TgHipSceneDesc *s, int bi, Event *e);
static float bsdf_pdf(const TgHipSceneDesc *s, int bi, const Event *e);

/* RoughDielectricBsdf::sampleBase/evalBase/pdfBase (RoughDielectricBsdf.cpp:55-131, 133-166, 200-236) */
static int rd_sampleBase(Event *e, int sampleR, int sampleT, float roughness, float ior, int dist)
{
    float wiDotN = e->wi.z;
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    float sampleRoughness = (1.2f - 0.2f*sqrtf(fabsf(wiDotN)))*roughness;
    float alpha = mf_roughnessToAlpha(dist, roughness);
    float sampleAlpha = mf_roughnessToAlpha(dist, sampleRoughness);

    float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
    v3 m = mf_sample(dist, sampleAlpha, xi0, xi1);
    float pm = mf_pdf(dist, sampleAlpha, m);
    if (pm < 1e-10f)
        return 0;

    float wiDotM = vdot(e->wi, m);
    float cosThetaT = 0.0f;
    float F = dielectricReflectanceT(1.0f/ior, wiDotM, &cosThetaT);
    float etaM = wiDotM < 0.0f ? ior : 1.0f/ior;

    int reflect;
    if (sampleR && sampleT) {
        reflect = nextBoolean(e->sampler, F);
    } else if (sampleT) {
        if (F == 1.0f)
            return 0;
        reflect = 0;
    } else if (sampleR) {
        reflect = 1;
    } else {
        return 0;
    }

    if (reflect)
        e->wo = vsub(vscale(m, 2.0f*wiDotM), e->wi);
    else
        e->wo = vsub(vscale(m, etaM*wiDotM - sgnE(wiDotM)*cosThetaT), vscale(e->wi, etaM));

    float woDotN = e->wo.z;
    int reflected = wiDotN*woDotN > 0.0f;
    if (reflected != reflect)
        return 0;

    float woDotM = vdot(e->wo, m);
    float G = mf_G(dist, alpha, e->wi, e->wo, m);
    float D = mf_D(dist, alpha, m);
    e->weight = vs(fabsf(wiDotM)*G*D/(fabsf(wiDotN)*pm));

    if (reflect) {
        e->pdf = pm*0.25f/fabsf(wiDotM);
        e->sampled = TGHIP_LOBE_GLOSSY_R;
    } else {
        e->pdf = pm*fabsf(woDotM)/sqr(eta*wiDotM + woDotM);
        e->sampled = TGHIP_LOBE_GLOSSY_T;
    }
    if (sampleR && sampleT) {
        if (reflect) e->pdf *= F;
        else e->pdf *= 1.0f - F;
    } else {
        if (reflect) e->weight = vscale(e->weight, F);
        else e->weight = vscale(e->weight, 1.0f - F);
    }
    return 1;
}
static v3 rd_evalBase(const Event *e, int sampleR, int sampleT, float roughness, float ior, int dist)
{
    float wiDotN = e->wi.z, woDotN = e->wo.z;
    int reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT))
        return vs(0.0f);
    float alpha = mf_roughnessToAlpha(dist, roughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    v3 m;
    if (reflect)
        m = vscale(vnorm(vadd(e->wi, e->wo)), sgnE(wiDotN));
    else
        m = vneg(vnorm(vadd(vscale(e->wi, eta), e->wo)));
    float wiDotM = vdot(e->wi, m), woDotM = vdot(e->wo, m);
    float F = dielectricReflectance(1.0f/ior, wiDotM);
    float G = mf_G(dist, alpha, e->wi, e->wo, m);
    float D = mf_D(dist, alpha, m);
    if (reflect) {
        float fr = (F*G*D*0.25f)/fabsf(wiDotN);
        return vs(fr);
    } else {
        float fs = fabsf(wiDotM*woDotM)*(1.0f - F)*G*D/(sqr(eta*wiDotM + woDotM)*fabsf(wiDotN));
        return vs(fs);
    }
}
static float rd_pdfBase(const Event *e, int sampleR, int sampleT, float roughness, float ior, int dist)
{
    float wiDotN = e->wi.z, woDotN = e->wo.z;
    int reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT))
        return 0.0f;
    float sampleRoughness = (1.2f - 0.2f*sqrtf(fabsf(wiDotN)))*roughness;
    float sampleAlpha = mf_roughnessToAlpha(dist, sampleRoughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    v3 m;
    if (reflect)
        m = vscale(vnorm(vadd(e->wi, e->wo)), sgnE(wiDotN));
    else
        m = vneg(vnorm(vadd(vscale(e->wi, eta), e->wo)));
    float wiDotM = vdot(e->wi, m), woDotM = vdot(e->wo, m);
    float F = dielectricReflectance(1.0f/ior, wiDotM);
    float pm = mf_pdf(dist, sampleAlpha, m);
    float pdf;
    if (reflect)
        pdf = pm*0.25f/fabsf(wiDotM);
    else
        pdf = pm*fabsf(woDotM)/sqr(eta*wiDotM + woDotM);
    if (sampleR && sampleT) {
        if (reflect) pdf *= F;
        else pdf *= 1.0f - F;
    }
    return pdf;
}

/* diffuse substrate term shared by Plastic / RoughPlastic */
static v3 plastic_substrate(const TgHipBsdf *b, v3 diffuseAlbedo)
{
    v3 denom = vsub(vs(1.0f), vscale(diffuseAlbedo, b->diffuse_fresnel));
    return vdiv(diffuseAlbedo, denom);
}

static v3 bsdf_eval(const TgHipSceneDesc *s, int bi, const Event *e)
{
    const TgHipBsdf *b = &s->bsdfs[bi];
    switch (b->type) {
    case TGHIP_BSDF_LAMBERT:   /* LambertBsdf.cpp:40-47 */
    case TGHIP_BSDF_ERROR:
        if (!(e->requested & TGHIP_LOBE_DIFFUSE_R)) return vs(0.0f);
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        return vscale(vscale(bsdf_albedo(s, b, e), O_INV_PI), e->wo.z);
    case TGHIP_BSDF_NULL:      /* NullBsdf.cpp:24-27 */
        return vs(0.0f);
    case TGHIP_BSDF_FORWARD:   /* ForwardBsdf.cpp:25-28 */
        return (e->requested == TGHIP_LOBE_FORWARD && -e->wi.x == e->wo.x && -e->wi.y == e->wo.y && -e->wi.z == e->wo.z)
            ? vs(1.0f) : vs(0.0f);
    case TGHIP_BSDF_MIRROR:    /* MirrorBsdf.cpp:39-46 */
        if ((e->requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e->wi, e->wo))
            return bsdf_albedo(s, b, e);
        return vs(0.0f);
    case TGHIP_BSDF_CONDUCTOR: /* ConductorBsdf.cpp:68-75 */
        if ((e->requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e->wi, e->wo))
            return vmul(bsdf_albedo(s, b, e), conductorReflectance(b->eta, b->k, e->wi.z));
        return vs(0.0f);
    case TGHIP_BSDF_ROUGH_CONDUCTOR: { /* RoughConductorBsdf.cpp:93-109 */
        if (!(e->requested & TGHIP_LOBE_GLOSSY_R)) return vs(0.0f);
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        float roughness = bsdf_roughness(s, b, e);
        float alpha = mf_roughnessToAlpha(b->distribution, roughness);
        v3 hr = vnorm(vadd(e->wi, e->wo));
        float cosThetaM = vdot(e->wi, hr);
        v3 F = conductorReflectance(b->eta, b->k, cosThetaM);
        float G = mf_G(b->distribution, alpha, e->wi, e->wo, hr);
        float D = mf_D(b->distribution, alpha, hr);
        float fr = (G*D*0.25f)/e->wi.z;
        return vmul(bsdf_albedo(s, b, e), vscale(F, fr));
    }
    case TGHIP_BSDF_SMOOTH_COAT: { /* SmoothCoatBsdf.cpp:146-177 */
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        int evalR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int evalT = (e->requested & s->bsdfs[b->sub0].lobes) != 0;
        v3 wi = e->wi, wo = e->wo;
        float eta = 1.0f/b->ior;
        float cosThetaTi, cosThetaTo;
        float Fi = dielectricReflectanceT(eta, wi.z, &cosThetaTi);
        float Fo = dielectricReflectanceT(eta, wo.z, &cosThetaTo);
        if (evalR && checkReflectionConstraint(wi, wo)) {
            return vs(Fi);
        } else if (evalT) {
            Event q = *e;
            q.wi = V(wi.x*eta, wi.y*eta, copysignf(cosThetaTi, wi.z));
            q.wo = V(wo.x*eta, wo.y*eta, copysignf(cosThetaTo, wo.z));
            float laplacian = eta*eta*wo.z/cosThetaTo;
            v3 substrateF = bsdf_eval(s, b->sub0, &q);
            v3 ssa = ld3(b->scaled_sigma_a);
            if (vmax3(ssa) > 0.0f)
                substrateF = vmul(substrateF, vexp(vscale(ssa, -1.0f/cosThetaTo - 1.0f/cosThetaTi)));
            return vscale(substrateF, laplacian*(1.0f - Fi)*(1.0f - Fo));
        }
        return vs(0.0f);
    }
    case TGHIP_BSDF_DIELECTRIC: { /* DielectricBsdf.cpp:88-108 */
        int evalR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int evalT = (e->requested & TGHIP_LOBE_SPECULAR_T) && b->enable_refraction;
        float eta = e->wi.z < 0.0f ? b->ior : 1.0f/b->ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectanceT(eta, fabsf(e->wi.z), &cosThetaT);
        if (e->wi.z*e->wo.z >= 0.0f) {
            if (evalR && checkReflectionConstraint(e->wi, e->wo))
                return vscale(bsdf_albedo(s, b, e), F);
            return vs(0.0f);
        } else {
            if (evalT && checkRefractionConstraint(e->wi, e->wo, eta, cosThetaT))
                return vscale(bsdf_albedo(s, b, e), 1.0f - F);
            return vs(0.0f);
        }
    }
    case TGHIP_BSDF_ROUGH_DIELECTRIC: { /* RoughDielectricBsdf.cpp:247-254 */
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_GLOSSY_T) && b->enable_refraction;
        float roughness = bsdf_roughness(s, b, e);
        return vmul(rd_evalBase(e, sampleR, sampleT, roughness, b->ior, b->distribution), bsdf_albedo(s, b, e));
    }
    case TGHIP_BSDF_PLASTIC: { /* PlasticBsdf.cpp:125-151 */
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        int evalR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int evalT = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        float eta = 1.0f/b->ior;
        float Fi = dielectricReflectance(eta, e->wi.z);
        float Fo = dielectricReflectance(eta, e->wo.z);
        if (evalR && checkReflectionConstraint(e->wi, e->wo)) {
            return vs(Fi);
        } else if (evalT) {
            v3 diffuseAlbedo = bsdf_albedo(s, b, e);
            v3 brdf = vscale(plastic_substrate(b, diffuseAlbedo), (1.0f - Fi)*(1.0f - Fo)*eta*eta*e->wo.z*O_INV_PI);
            v3 ssa = ld3(b->scaled_sigma_a);
            if (vmax3(ssa) > 0.0f)
                brdf = vmul(brdf, vexp(vscale(ssa, -1.0f/e->wo.z - 1.0f/e->wi.z)));
            return brdf;
        }
        return vs(0.0f);
    }
    case TGHIP_BSDF_ROUGH_PLASTIC: { /* RoughPlasticBsdf.cpp:114-141 */
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        if (!sampleR && !sampleT) return vs(0.0f);
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        v3 glossyR = vs(0.0f);
        if (sampleR)
            glossyR = rd_evalBase(e, 1, 0, bsdf_roughness(s, b, e), b->ior, b->distribution);
        v3 diffuseR = vs(0.0f);
        if (sampleT) {
            float eta = 1.0f/b->ior;
            float Fi = dielectricReflectance(eta, e->wi.z);
            float Fo = dielectricReflectance(eta, e->wo.z);
            v3 diffuseAlbedo = bsdf_albedo(s, b, e);
            diffuseR = vscale(plastic_substrate(b, diffuseAlbedo), (1.0f - Fi)*(1.0f - Fo)*eta*eta*e->wo.z*O_INV_PI);
            v3 ssa = ld3(b->scaled_sigma_a);
            if (vmax3(ssa) > 0.0f)
                diffuseR = vmul(diffuseR, vexp(vscale(ssa, -1.0f/e->wo.z - 1.0f/e->wi.z)));
        }
        return vadd(glossyR, diffuseR);
    }
    case TGHIP_BSDF_MIXED: { /* MixedBsdf.cpp:101-105 */
        float ratio = texture_eval(s, b->tex1, e->u, e->v).x;
        v3 f0 = bsdf_eval(s, b->sub0, e), f1 = bsdf_eval(s, b->sub1, e);
        return vmul(bsdf_albedo(s, b, e), vadd(vscale(f0, ratio), vscale(f1, 1.0f - ratio)));
    }
    case TGHIP_BSDF_DIFFUSE_TRANSMISSION: { /* DiffuseTransmissionBsdf.cpp:50-57 */
        if (!(e->requested & TGHIP_LOBE_DIFFUSE_T)) return vs(0.0f);
        float factor = e->wi.z*e->wo.z < 0.0f ? b->eta[0] : 1.0f - b->eta[0];
        return vscale(vscale(vscale(bsdf_albedo(s, b, e), factor), O_INV_PI), fabsf(e->wo.z));
    }
    case TGHIP_BSDF_PHONG: { /* PhongBsdf.cpp:79-99 */
        int evalGlossy = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0, evalDiffuse = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        if (!evalGlossy && !evalDiffuse) return vs(0.0f);
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        float result = 0.0f;
        if (evalDiffuse)
            result += b->eta[1]*O_INV_PI;
        if (evalGlossy) {
            float cosTheta = vdot(V(-e->wi.x, -e->wi.y, e->wi.z), e->wo);
            if (cosTheta > 0.0f)
                result += powf(cosTheta, b->eta[0])*b->k[2]*(1.0f - b->eta[1]);
        }
        return vscale(vscale(bsdf_albedo(s, b, e), e->wo.z), result);
    }
    case TGHIP_BSDF_THINSHEET: { /* ThinSheetBsdf.cpp:83-104 */
        if (e->requested != TGHIP_LOBE_FORWARD || -e->wi.x != e->wo.x || -e->wi.y != e->wo.y || -e->wi.z != e->wo.z)
            return vs(0.0f);
        float thickness = texture_eval(s, b->tex1, e->u, e->v).x;
        float cosThetaT;
        v3 transmittance;
        if (b->enable_refraction) {
            v3 r = thinFilmReflectanceInterference(1.0f/b->ior, fabsf(e->wi.z), thickness*500.0f, &cosThetaT);
            transmittance = V(1.0f - r.x, 1.0f - r.y, 1.0f - r.z);
        } else {
            transmittance = vs(1.0f - thinFilmReflectance(1.0f/b->ior, fabsf(e->wi.z), &cosThetaT));
        }
        v3 sa = ld3(b->sigma_a);
        if (!(sa.x == 0.0f && sa.y == 0.0f && sa.z == 0.0f) && cosThetaT > 0.0f)
            transmittance = vmul(transmittance, vexp(vscale(vneg(sa), thickness*2.0f/cosThetaT)));
        return transmittance;
    }
    case TGHIP_BSDF_OREN_NAYAR: { /* OrenNayarBsdf.cpp:61-100 */
        if (!(e->requested & TGHIP_LOBE_DIFFUSE_R)) return vs(0.0f);
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        v3 wi = e->wi, wo = e->wo;
        float thetaR = acosf(wo.z);
        float thetaI = acosf(wi.z);
        float alpha = thetaR > thetaI ? thetaR : thetaI;       /* Tungsten::max / min (a < b ? b : a) */
        float beta = thetaR < thetaI ? thetaR : thetaI;
        float sinAlpha = sinf(alpha);
        float denom = (wi.x*wi.x + wi.y*wi.y)*(wo.x*wo.x + wo.y*wo.y);
        float cosDeltaPhi;
        if (denom == 0.0f)
            cosDeltaPhi = 1.0f;
        else
            cosDeltaPhi = (wi.x*wo.x + wi.y*wo.y)/sqrtf(denom);
        const float RoughnessToSigma = 1.0f/sqrtf(2.0f);
        float sigma = RoughnessToSigma*bsdf_roughness(s, b, e);
        float sigmaSq = sigma*sigma;
        float C1 = 1.0f - 0.5f*sigmaSq/(sigmaSq + 0.33f);
        float C2 = 0.45f*sigmaSq/(sigmaSq + 0.09f);
        if (cosDeltaPhi >= 0.0f) {
            C2 *= sinAlpha;
        } else {
            float q = (2.0f*O_INV_PI)*beta;
            C2 *= sinAlpha - q*q*q;
        }
        float C3 = 0.125f*(sigmaSq/(sigmaSq + 0.09f))*sqr((4.0f*O_INV_PI*O_INV_PI)*alpha*beta);
        float fr1 = (C1 + cosDeltaPhi*C2*tanf(beta) + (1.0f - fabsf(cosDeltaPhi))*C3*tanf(0.5f*(alpha + beta)));
        float fr2 = 0.17f*sigmaSq/(sigmaSq + 0.13f)*(1.0f - cosDeltaPhi*sqr((2.0f*O_INV_PI)*beta));
        v3 diffuseAlbedo = bsdf_albedo(s, b, e);
        return vscale(vscale(vadd(vscale(diffuseAlbedo, fr1), vscale(vmul(diffuseAlbedo, diffuseAlbedo), fr2)), wo.z), O_INV_PI);
    }
    case TGHIP_BSDF_ROUGH_COAT: { /* RoughCoatBsdf.cpp:161-199 */
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & s->bsdfs[b->sub0].lobes) != 0;
        if (!sampleT && !sampleR) return vs(0.0f);
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return vs(0.0f);
        v3 glossyR = vs(0.0f);
        if (sampleR)
            glossyR = rd_evalBase(e, 1, 0, bsdf_roughness(s, b, e), b->ior, b->distribution);
        v3 substrateR = vs(0.0f);
        if (sampleT) {
            v3 wi = e->wi, wo = e->wo;
            float eta = 1.0f/b->ior;
            float cosThetaTi, cosThetaTo;
            float Fi = dielectricReflectanceT(eta, wi.z, &cosThetaTi);
            float Fo = dielectricReflectanceT(eta, wo.z, &cosThetaTo);
            if (Fi == 1.0f || Fo == 1.0f)
                return glossyR;
            Event q = *e;
            q.wi = V(wi.x*eta, wi.y*eta, copysignf(cosThetaTi, wi.z));
            q.wo = V(wo.x*eta, wo.y*eta, copysignf(cosThetaTo, wo.z));
            float compressionProjection = eta*eta*wo.z/cosThetaTo;
            v3 substrateF = bsdf_eval(s, b->sub0, &q);
            v3 ssa = ld3(b->scaled_sigma_a);
            if (vmax3(ssa) > 0.0f)
                substrateF = vmul(substrateF, vexp(vscale(ssa, -1.0f/cosThetaTo - 1.0f/cosThetaTi)));
            substrateR = vscale(substrateF, compressionProjection*(1.0f - Fi)*(1.0f - Fo));
        }
        return vadd(glossyR, substrateR);
    }
    case TGHIP_BSDF_TRANSPARENCY: /* TransparencyBsdf.cpp:48-54 */
        if (e->requested == TGHIP_LOBE_FORWARD)
            return (-e->wi.x == e->wo.x && -e->wi.y == e->wo.y && -e->wi.z == e->wo.z)
                ? vs(1.0f - texture_eval(s, b->tex1, e->u, e->v).x) : vs(0.0f);
        return bsdf_eval(s, b->sub0, e);
    }
    return vs(0.0f);
}

static int mixed_adjustedRatio(const TgHipSceneDesc *s, const TgHipBsdf *b, const Event *e, float *ratio)   /* MixedBsdf.cpp:17-31 */
{
    int sample0 = (e->requested & s->bsdfs[b->sub0].lobes) != 0;
    int sample1 = (e->requested & s->bsdfs[b->sub1].lobes) != 0;
    if (sample0 && sample1) *ratio = texture_eval(s, b->tex1, e->u, e->v).x;
    else if (sample0) *ratio = 1.0f;
    else if (sample1) *ratio = 0.0f;
    else return 0;
    return 1;
}

static int bsdf_sample(const TgHipSceneDesc *s, int bi, Event *e)
{
    const TgHipBsdf *b = &s->bsdfs[bi];
    switch (b->type) {
    case TGHIP_BSDF_LAMBERT:   /* LambertBsdf.cpp:27-38 */
    case TGHIP_BSDF_ERROR: {
        if (!(e->requested & TGHIP_LOBE_DIFFUSE_R)) return 0;
        if (e->wi.z <= 0.0f) return 0;
        float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
        e->wo = cosineHemisphere(xi0, xi1);
        e->pdf = cosineHemispherePdf(e->wo);
        e->weight = bsdf_albedo(s, b, e);
        e->sampled = TGHIP_LOBE_DIFFUSE_R;
        return 1;
    }
    case TGHIP_BSDF_NULL:
    case TGHIP_BSDF_FORWARD:
        return 0;
    case TGHIP_BSDF_MIRROR:    /* MirrorBsdf.cpp:28-37 */
        if (!(e->requested & TGHIP_LOBE_SPECULAR_R)) return 0;
        e->wo = V(-e->wi.x, -e->wi.y, e->wi.z);
        e->pdf = 1.0f;
        e->sampled = TGHIP_LOBE_SPECULAR_R;
        e->weight = bsdf_albedo(s, b, e);
        return 1;
    case TGHIP_BSDF_CONDUCTOR: /* ConductorBsdf.cpp:56-66 */
        if (!(e->requested & TGHIP_LOBE_SPECULAR_R)) return 0;
        e->wo = V(-e->wi.x, -e->wi.y, e->wi.z);
        e->pdf = 1.0f;
        e->weight = vmul(bsdf_albedo(s, b, e), conductorReflectance(b->eta, b->k, e->wi.z));
        e->sampled = TGHIP_LOBE_SPECULAR_R;
        return 1;
    case TGHIP_BSDF_ROUGH_CONDUCTOR: { /* RoughConductorBsdf.cpp:60-91 */
        if (!(e->requested & TGHIP_LOBE_GLOSSY_R)) return 0;
        if (e->wi.z <= 0.0f) return 0;
        float roughness = bsdf_roughness(s, b, e);
        float sampleRoughness = roughness;
        float alpha = mf_roughnessToAlpha(b->distribution, roughness);
        float sampleAlpha = mf_roughnessToAlpha(b->distribution, sampleRoughness);
        float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
        v3 m = mf_sample(b->distribution, sampleAlpha, xi0, xi1);
        float wiDotM = vdot(e->wi, m);
        e->wo = vsub(vscale(m, 2.0f*wiDotM), e->wi);
        if (wiDotM <= 0.0f || e->wo.z <= 0.0f)
            return 0;
        float G = mf_G(b->distribution, alpha, e->wi, e->wo, m);
        float D = mf_D(b->distribution, alpha, m);
        float mPdf = mf_pdf(b->distribution, sampleAlpha, m);
        float pdf = mPdf*0.25f/wiDotM;
        float weight = wiDotM*G*D/(e->wi.z*mPdf);
        v3 F = conductorReflectance(b->eta, b->k, wiDotM);
        e->pdf = pdf;
        e->weight = vmul(bsdf_albedo(s, b, e), vscale(F, weight));
        e->sampled = TGHIP_LOBE_GLOSSY_R;
        return 1;
    }
    case TGHIP_BSDF_SMOOTH_COAT: { /* SmoothCoatBsdf.cpp:41-100 */
        if (e->wi.z <= 0.0f) return 0;
        int sampleR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int sampleT = (e->requested & s->bsdfs[b->sub0].lobes) != 0;
        if (!sampleR && !sampleT) return 0;
        v3 wi = e->wi;
        float eta = 1.0f/b->ior;
        float cosThetaTi;
        float Fi = dielectricReflectanceT(eta, wi.z, &cosThetaTi);
        float substrateWeight = b->avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability;
        if (sampleR && sampleT) specularProbability = specularWeight/(specularWeight + substrateWeight);
        else if (sampleR) specularProbability = 1.0f;
        else specularProbability = 0.0f;

        if (sampleR && nextBoolean(e->sampler, specularProbability)) {
            e->wo = V(-wi.x, -wi.y, wi.z);
            e->pdf = specularProbability;
            e->weight = vs(Fi/specularProbability);
            e->sampled = TGHIP_LOBE_SPECULAR_R;
        } else {
            v3 originalWi = wi;
            e->wi = V(wi.x*eta, wi.y*eta, cosThetaTi);
            int success = bsdf_sample(s, b->sub0, e);
            e->wi = originalWi;
            if (!success) return 0;
            float cosThetaTo;
            float Fo = dielectricReflectanceT(b->ior, e->wo.z, &cosThetaTo);
            if (Fo == 1.0f) return 0;
            float cosThetaSubstrate = e->wo.z;
            e->wo = V(e->wo.x*b->ior, e->wo.y*b->ior, cosThetaTo);
            e->weight = vscale(e->weight, (1.0f - Fi)*(1.0f - Fo));
            v3 ssa = ld3(b->scaled_sigma_a);
            if (vmax3(ssa) > 0.0f)
                e->weight = vmul(e->weight, vexp(vscale(ssa, -1.0f/cosThetaSubstrate - 1.0f/cosThetaTi)));
            e->weight = vdivs(e->weight, 1.0f - specularProbability);
            e->pdf *= 1.0f - specularProbability;
            e->pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
        }
        return 1;
    }
    case TGHIP_BSDF_DIELECTRIC: { /* DielectricBsdf.cpp:49-86 */
        int sampleR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_SPECULAR_T) && b->enable_refraction;
        float eta = e->wi.z < 0.0f ? b->ior : 1.0f/b->ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectanceT(eta, fabsf(e->wi.z), &cosThetaT);
        float reflectionProbability;
        if (sampleR && sampleT) reflectionProbability = F;
        else if (sampleR) reflectionProbability = 1.0f;
        else if (sampleT) reflectionProbability = 0.0f;
        else return 0;
        if (nextBoolean(e->sampler, reflectionProbability)) {
            e->wo = V(-e->wi.x, -e->wi.y, e->wi.z);
            e->pdf = reflectionProbability;
            e->sampled = TGHIP_LOBE_SPECULAR_R;
            e->weight = sampleT ? vs(1.0f) : vs(F);
        } else {
            if (F == 1.0f) return 0;
            e->wo = V(-e->wi.x*eta, -e->wi.y*eta, -copysignf(cosThetaT, e->wi.z));
            e->pdf = 1.0f - reflectionProbability;
            e->sampled = TGHIP_LOBE_SPECULAR_T;
            e->weight = sampleR ? vs(1.0f) : vs(1.0f - F);
        }
        e->weight = vmul(e->weight, bsdf_albedo(s, b, e));
        return 1;
    }
    case TGHIP_BSDF_ROUGH_DIELECTRIC: { /* RoughDielectricBsdf.cpp:238-245 */
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_GLOSSY_T) && b->enable_refraction;
        float roughness = bsdf_roughness(s, b, e);
        int result = rd_sampleBase(e, sampleR, sampleT, roughness, b->ior, b->distribution);
        e->weight = vmul(e->weight, bsdf_albedo(s, b, e));
        return result;
    }
    case TGHIP_BSDF_PLASTIC: { /* PlasticBsdf.cpp:45-87 */
        if (e->wi.z <= 0.0f) return 0;
        int sampleR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        v3 wi = e->wi;
        float eta = 1.0f/b->ior;
        float Fi = dielectricReflectance(eta, wi.z);
        float substrateWeight = b->avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability;
        if (sampleR && sampleT) specularProbability = specularWeight/(specularWeight + substrateWeight);
        else if (sampleR) specularProbability = 1.0f;
        else if (sampleT) specularProbability = 0.0f;
        else return 0;
        if (sampleR && nextBoolean(e->sampler, specularProbability)) {
            e->wo = V(-wi.x, -wi.y, wi.z);
            e->pdf = specularProbability;
            e->weight = vs(Fi/specularProbability);
            e->sampled = TGHIP_LOBE_SPECULAR_R;
        } else {
            float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
            v3 wo = cosineHemisphere(xi0, xi1);
            float Fo = dielectricReflectance(eta, wo.z);
            v3 diffuseAlbedo = bsdf_albedo(s, b, e);
            e->wo = wo;
            e->weight = vscale(plastic_substrate(b, diffuseAlbedo), (1.0f - Fi)*(1.0f - Fo)*eta*eta);
            v3 ssa = ld3(b->scaled_sigma_a);
            if (vmax3(ssa) > 0.0f)
                e->weight = vmul(e->weight, vexp(vscale(ssa, -1.0f/e->wo.z - 1.0f/e->wi.z)));
            e->pdf = cosineHemispherePdf(e->wo)*(1.0f - specularProbability);
            e->weight = vdivs(e->weight, 1.0f - specularProbability);
            e->sampled = TGHIP_LOBE_DIFFUSE_R;
        }
        return 1;
    }
    case TGHIP_BSDF_ROUGH_PLASTIC: { /* RoughPlasticBsdf.cpp:54-112 */
        if (e->wi.z <= 0.0f) return 0;
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        if (!sampleR && !sampleT) return 0;
        v3 wi = e->wi;
        float eta = 1.0f/b->ior;
        float Fi = dielectricReflectance(eta, wi.z);
        float substrateW = vavg(ld3(s->textures[b->albedo].avg));     /* _substrateWeight = _albedo->average().avg() */
        float substrateWeight = substrateW*b->avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability = specularWeight/(specularWeight + substrateWeight);
        if (sampleR && (nextBoolean(e->sampler, specularProbability) || !sampleT)) {
            float roughness = bsdf_roughness(s, b, e);
            if (!rd_sampleBase(e, 1, 0, roughness, b->ior, b->distribution))
                return 0;
            if (sampleT) {
                v3 diffuseAlbedo = bsdf_albedo(s, b, e);
                float Fo = dielectricReflectance(eta, e->wo.z);
                v3 brdfSubstrate = vscale(vscale(vscale(plastic_substrate(b, diffuseAlbedo), (1.0f - Fi)*(1.0f - Fo)*eta*eta), O_INV_PI), e->wo.z);
                v3 brdfSpecular = vscale(e->weight, e->pdf);
                float pdfSubstrate = cosineHemispherePdf(e->wo)*(1.0f - specularProbability);
                float pdfSpecular = e->pdf*specularProbability;
                e->weight = vdivs(vadd(brdfSpecular, brdfSubstrate), pdfSpecular + pdfSubstrate);
                e->pdf = pdfSpecular + pdfSubstrate;
            }
            return 1;
        } else {
            float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
            v3 wo = cosineHemisphere(xi0, xi1);
            float Fo = dielectricReflectance(eta, wo.z);
            v3 diffuseAlbedo = bsdf_albedo(s, b, e);
            e->wo = wo;
            e->weight = vscale(plastic_substrate(b, diffuseAlbedo), (1.0f - Fi)*(1.0f - Fo)*eta*eta);
            v3 ssa = ld3(b->scaled_sigma_a);
            if (vmax3(ssa) > 0.0f)
                e->weight = vmul(e->weight, vexp(vscale(ssa, -1.0f/e->wo.z - 1.0f/e->wi.z)));
            e->pdf = cosineHemispherePdf(e->wo);
            if (sampleR) {
                v3 brdfSubstrate = vscale(e->weight, e->pdf);
                float pdfSubstrate = e->pdf*(1.0f - specularProbability);
                float r = bsdf_roughness(s, b, e);
                v3 brdfSpecular = rd_evalBase(e, 1, 0, r, b->ior, b->distribution);
                float pdfSpecular = rd_pdfBase(e, 1, 0, r, b->ior, b->distribution);
                pdfSpecular *= specularProbability;
                e->weight = vdivs(vadd(brdfSpecular, brdfSubstrate), pdfSpecular + pdfSubstrate);
                e->pdf = pdfSpecular + pdfSubstrate;
            }
            e->sampled = TGHIP_LOBE_DIFFUSE_R;
        }
        return 1;
    }
    case TGHIP_BSDF_MIXED: { /* MixedBsdf.cpp:70-99 */
        float ratio;
        if (!mixed_adjustedRatio(s, b, e, &ratio)) return 0;
        if (nextBoolean(e->sampler, ratio)) {
            if (!bsdf_sample(s, b->sub0, e)) return 0;
            float pdf0 = e->pdf*ratio;
            float pdf1 = bsdf_pdf(s, b->sub1, e)*(1.0f - ratio);
            v3 f = vadd(vscale(vscale(e->weight, e->pdf), ratio), vscale(bsdf_eval(s, b->sub1, e), 1.0f - ratio));
            e->pdf = pdf0 + pdf1;
            e->weight = vdivs(f, e->pdf);
        } else {
            if (!bsdf_sample(s, b->sub1, e)) return 0;
            float pdf0 = bsdf_pdf(s, b->sub0, e)*ratio;
            float pdf1 = e->pdf*(1.0f - ratio);
            v3 f = vadd(vscale(bsdf_eval(s, b->sub0, e), ratio), vscale(vscale(e->weight, e->pdf), 1.0f - ratio));
            e->pdf = pdf0 + pdf1;
            e->weight = vdivs(f, e->pdf);
        }
        e->weight = vmul(e->weight, bsdf_albedo(s, b, e));
        return 1;
    }
    case TGHIP_BSDF_DIFFUSE_TRANSMISSION: { /* DiffuseTransmissionBsdf.cpp:29-48 */
        int sampleR = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0, sampleT = (e->requested & TGHIP_LOBE_DIFFUSE_T) != 0;
        if (!sampleR && !sampleT) return 0;
        const float T = b->eta[0];
        float transmittanceProbability = sampleR && sampleT ? T : (sampleR ? 0.0f : 1.0f);
        int transmit = nextBoolean(e->sampler, transmittanceProbability);
        float weight = sampleR && sampleT ? 1.0f : (transmit ? T : 1.0f - T);
        float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
        e->wo = cosineHemisphere(xi0, xi1);
        e->wo.z = copysignf(e->wo.z, e->wi.z);
        if (transmit)
            e->wo.z = -e->wo.z;
        e->pdf = cosineHemispherePdf(e->wo);
        e->weight = vscale(bsdf_albedo(s, b, e), weight);
        e->sampled = TGHIP_LOBE_DIFFUSE_T;
        return 1;
    }
    case TGHIP_BSDF_PHONG: { /* PhongBsdf.cpp:39-77 */
        int evalGlossy = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0, evalDiffuse = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        if (!evalGlossy && !evalDiffuse) return 0;
        if (e->wi.z <= 0.0f) return 0;
        int sampleGlossy;
        if (evalGlossy && evalDiffuse)
            sampleGlossy = nextBoolean(e->sampler, 1.0f - b->eta[1]);
        else
            sampleGlossy = evalGlossy;
        if (sampleGlossy) {
            float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
            float phi = xi0*O_TWO_PI;
            float cosTheta = powf(xi1, b->k[0]);
            float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta*cosTheta));
            v3 woLocal = V(cosf(phi)*sinTheta, sinf(phi)*sinTheta, cosTheta);
            Frame lobe = frame_from_normal(V(-e->wi.x, -e->wi.y, e->wi.z));
            e->wo = toGlobal(&lobe, woLocal);
            if (e->wo.z < 0.0f)
                return 0;
            e->sampled = TGHIP_LOBE_GLOSSY_R;
        } else {
            float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
            e->wo = cosineHemisphere(xi0, xi1);
            e->sampled = TGHIP_LOBE_DIFFUSE_R;
        }
        e->pdf = bsdf_pdf(s, bi, e);
        e->weight = vdivs(bsdf_eval(s, bi, e), e->pdf);
        return 1;
    }
    case TGHIP_BSDF_THINSHEET: { /* ThinSheetBsdf.cpp:49-81 */
        if (!(e->requested & TGHIP_LOBE_SPECULAR_R)) return 0;
        e->wo = V(-e->wi.x, -e->wi.y, e->wi.z);
        e->pdf = 1.0f;
        e->sampled = TGHIP_LOBE_SPECULAR_R;
        v3 sa = ld3(b->sigma_a);
        const int absorbing = !(sa.x == 0.0f && sa.y == 0.0f && sa.z == 0.0f);
        if (!absorbing && !b->enable_refraction) {
            e->weight = vs(1.0f);                           /* "fast path / early out" */
            return 1;
        }
        float thickness = texture_eval(s, b->tex1, e->u, e->v).x;
        float cosThetaT;
        if (b->enable_refraction)
            e->weight = thinFilmReflectanceInterference(1.0f/b->ior, fabsf(e->wi.z), thickness*500.0f, &cosThetaT);
        else
            e->weight = vs(thinFilmReflectance(1.0f/b->ior, fabsf(e->wi.z), &cosThetaT));
        v3 transmittance = V(1.0f - e->weight.x, 1.0f - e->weight.y, 1.0f - e->weight.z);
        if (absorbing && cosThetaT > 0.0f)
            transmittance = vmul(transmittance, vexp(vscale(vneg(sa), thickness*2.0f/cosThetaT)));
        e->weight = vdivs(e->weight, 1.0f - vavg(transmittance));
        return 1;
    }
    case TGHIP_BSDF_OREN_NAYAR: { /* OrenNayarBsdf.cpp:41-59 */
        if (!(e->requested & TGHIP_LOBE_DIFFUSE_R)) return 0;
        if (e->wi.z <= 0.0f) return 0;
        float roughness = bsdf_roughness(s, b, e);
        float ratio = fminf(fmaxf(roughness, 0.01f), 1.0f);       /* clamp(val, 0.01f, 1.0f) */
        if (nextBoolean(e->sampler, ratio)) {
            float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
            e->wo = uniformHemisphere(xi0, xi1);
        } else {
            float xi0 = next1D(e->sampler), xi1 = next1D(e->sampler);
            e->wo = cosineHemisphere(xi0, xi1);
        }
        e->pdf = O_INV_TWO_PI*ratio + cosineHemispherePdf(e->wo)*(1.0f - ratio);
        e->weight = vdivs(bsdf_eval(s, bi, e), e->pdf);
        e->sampled = TGHIP_LOBE_DIFFUSE_R;
        return e->wo.z > 0.0f;
    }
    case TGHIP_BSDF_ROUGH_COAT: { /* RoughCoatBsdf.cpp:82-159 */
        if (e->wi.z <= 0.0f) return 0;
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & s->bsdfs[b->sub0].lobes) != 0;
        if (!sampleR && !sampleT) return 0;
        v3 wi = e->wi;
        float eta = 1.0f/b->ior;
        float cosThetaTi;
        float Fi = dielectricReflectanceT(eta, wi.z, &cosThetaTi);
        float substrateWeight = b->avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability = specularWeight/(specularWeight + substrateWeight);
        v3 ssa = ld3(b->scaled_sigma_a);
        if (sampleR && (nextBoolean(e->sampler, specularProbability) || !sampleT)) {
            float roughness = bsdf_roughness(s, b, e);
            if (!rd_sampleBase(e, 1, 0, roughness, b->ior, b->distribution))
                return 0;
            if (sampleT) {
                v3 brdfSpecular = vscale(e->weight, e->pdf);
                float pdfSpecular = e->pdf*specularProbability;
                /* substrateEvalAndPdf (:58-80) */
                v3 brdfSubstrate; float pdfSubstrate;
                float cosThetaTo;
                float Fo = dielectricReflectanceT(eta, e->wo.z, &cosThetaTo);
                if (Fi == 1.0f || Fo == 1.0f) {
                    pdfSubstrate = 0.0f;
                    brdfSubstrate = vs(0.0f);
                } else {
                    Event q = *e;
                    q.wi = V(wi.x*eta, wi.y*eta, copysignf(cosThetaTi, wi.z));
                    q.wo = V(e->wo.x*eta, e->wo.y*eta, copysignf(cosThetaTo, e->wo.z));
                    pdfSubstrate = bsdf_pdf(s, b->sub0, &q);
                    pdfSubstrate *= eta*eta*fabsf(e->wo.z/cosThetaTo);
                    float compressionProjection = eta*eta*e->wo.z/cosThetaTo;
                    v3 substrateF = bsdf_eval(s, b->sub0, &q);
                    if (vmax3(ssa) > 0.0f)
                        substrateF = vmul(substrateF, vexp(vscale(ssa, -1.0f/cosThetaTo - 1.0f/cosThetaTi)));
                    brdfSubstrate = vscale(substrateF, compressionProjection*(1.0f - Fi)*(1.0f - Fo));
                }
                pdfSubstrate *= 1.0f - specularProbability;
                e->weight = vdivs(vadd(brdfSpecular, brdfSubstrate), pdfSpecular + pdfSubstrate);
                e->pdf = pdfSpecular + pdfSubstrate;
            }
            return 1;
        } else {
            v3 originalWi = wi;
            v3 wiSubstrate = V(wi.x*eta, wi.y*eta, cosThetaTi);
            e->wi = wiSubstrate;
            int success = bsdf_sample(s, b->sub0, e);
            e->wi = originalWi;
            if (!success) return 0;
            float cosThetaTo;
            float Fo = dielectricReflectanceT(b->ior, e->wo.z, &cosThetaTo);
            if (Fo == 1.0f) return 0;
            float cosThetaSubstrate = e->wo.z;
            e->wo = V(e->wo.x*b->ior, e->wo.y*b->ior, cosThetaTo);
            e->weight = vscale(e->weight, (1.0f - Fi)*(1.0f - Fo));
            if (vmax3(ssa) > 0.0f)
                e->weight = vmul(e->weight, vexp(vscale(ssa, -1.0f/cosThetaSubstrate - 1.0f/cosThetaTi)));
            e->weight = vscale(e->weight, originalWi.z/wiSubstrate.z);
            e->pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
            if (sampleR) {
                v3 brdfSubstrate = vscale(e->weight, e->pdf);
                float pdfSubstrate = e->pdf*(1.0f - specularProbability);
                float r = bsdf_roughness(s, b, e);
                v3 brdfSpecular = rd_evalBase(e, 1, 0, r, b->ior, b->distribution);
                float pdfSpecular = rd_pdfBase(e, 1, 0, r, b->ior, b->distribution);
                pdfSpecular *= specularProbability;
                e->weight = vdivs(vadd(brdfSpecular, brdfSubstrate), pdfSpecular + pdfSubstrate);
                e->pdf = pdfSpecular + pdfSubstrate;
            }
        }
        return 1;
    }
    case TGHIP_BSDF_TRANSPARENCY: /* TransparencyBsdf.cpp:43-46 */
        return bsdf_sample(s, b->sub0, e);
    }
    return 0;
}

static float bsdf_pdf(const TgHipSceneDesc *s, int bi, const Event *e)
{
    const TgHipBsdf *b = &s->bsdfs[bi];
    switch (b->type) {
    case TGHIP_BSDF_LAMBERT:   /* LambertBsdf.cpp:61-68 */
    case TGHIP_BSDF_ERROR:
        if (!(e->requested & TGHIP_LOBE_DIFFUSE_R)) return 0.0f;
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        return cosineHemispherePdf(e->wo);
    case TGHIP_BSDF_NULL:
    case TGHIP_BSDF_FORWARD:
        return 0.0f;
    case TGHIP_BSDF_MIRROR:
    case TGHIP_BSDF_CONDUCTOR:  /* MirrorBsdf.cpp:57-64, ConductorBsdf.cpp:82-89 */
        return ((e->requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e->wi, e->wo)) ? 1.0f : 0.0f;
    case TGHIP_BSDF_ROUGH_CONDUCTOR: { /* RoughConductorBsdf.cpp:127-143 */
        if (!(e->requested & TGHIP_LOBE_GLOSSY_R)) return 0.0f;
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        float roughness = bsdf_roughness(s, b, e);
        float sampleAlpha = mf_roughnessToAlpha(b->distribution, roughness);
        v3 hr = vnorm(vadd(e->wi, e->wo));
        return mf_pdf(b->distribution, sampleAlpha, hr)*0.25f/vdot(e->wi, hr);
    }
    case TGHIP_BSDF_SMOOTH_COAT: { /* SmoothCoatBsdf.cpp:179-214 */
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        int sampleR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int sampleT = (e->requested & s->bsdfs[b->sub0].lobes) != 0;
        v3 wi = e->wi, wo = e->wo;
        float eta = 1.0f/b->ior;
        float cosThetaTi, cosThetaTo;
        float Fi = dielectricReflectanceT(eta, wi.z, &cosThetaTi);
        dielectricReflectanceT(eta, wo.z, &cosThetaTo);
        Event q = *e;
        q.wi = V(wi.x*eta, wi.y*eta, copysignf(cosThetaTi, wi.z));
        q.wo = V(wo.x*eta, wo.y*eta, copysignf(cosThetaTo, wo.z));
        if (sampleR && sampleT) {
            float substrateWeight = b->avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            float specularProbability = specularWeight/(specularWeight + substrateWeight);
            if (checkReflectionConstraint(wi, wo))
                return specularProbability;
            return bsdf_pdf(s, b->sub0, &q)*(1.0f - specularProbability)*eta*eta*fabsf(wo.z/cosThetaTo);
        } else if (sampleT) {
            return bsdf_pdf(s, b->sub0, &q)*eta*eta*fabsf(wo.z/cosThetaTo);
        } else if (sampleR) {
            return checkReflectionConstraint(wi, wo) ? 1.0f : 0.0f;
        }
        return 0.0f;
    }
    case TGHIP_BSDF_DIELECTRIC: { /* DielectricBsdf.cpp:143-164 */
        int sampleR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_SPECULAR_T) && b->enable_refraction;
        float eta = e->wi.z < 0.0f ? b->ior : 1.0f/b->ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectanceT(eta, fabsf(e->wi.z), &cosThetaT);
        if (e->wi.z*e->wo.z >= 0.0f) {
            if (sampleR && checkReflectionConstraint(e->wi, e->wo)) return sampleT ? F : 1.0f;
            return 0.0f;
        } else {
            if (sampleT && checkRefractionConstraint(e->wi, e->wo, eta, cosThetaT)) return sampleR ? 1.0f - F : 1.0f;
            return 0.0f;
        }
    }
    case TGHIP_BSDF_ROUGH_DIELECTRIC: { /* RoughDielectricBsdf.cpp:265-272 */
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_GLOSSY_T) && b->enable_refraction;
        return rd_pdfBase(e, sampleR, sampleT, bsdf_roughness(s, b, e), b->ior, b->distribution);
    }
    case TGHIP_BSDF_PLASTIC: { /* PlasticBsdf.cpp:153-177 */
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        int sampleR = (e->requested & TGHIP_LOBE_SPECULAR_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        if (sampleR && sampleT) {
            float Fi = dielectricReflectance(1.0f/b->ior, e->wi.z);
            float substrateWeight = b->avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            float specularProbability = specularWeight/(specularWeight + substrateWeight);
            if (checkReflectionConstraint(e->wi, e->wo)) return specularProbability;
            return cosineHemispherePdf(e->wo)*(1.0f - specularProbability);
        } else if (sampleT) {
            return cosineHemispherePdf(e->wo);
        } else if (sampleR) {
            return checkReflectionConstraint(e->wi, e->wo) ? 1.0f : 0.0f;
        }
        return 0.0f;
    }
    case TGHIP_BSDF_ROUGH_PLASTIC: { /* RoughPlasticBsdf.cpp:185-213 */
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        if (!sampleR && !sampleT) return 0.0f;
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        float glossyPdf = 0.0f;
        if (sampleR) glossyPdf = rd_pdfBase(e, 1, 0, bsdf_roughness(s, b, e), b->ior, b->distribution);
        float diffusePdf = 0.0f;
        if (sampleT) diffusePdf = cosineHemispherePdf(e->wo);
        if (sampleT && sampleR) {
            float Fi = dielectricReflectance(1.0f/b->ior, e->wi.z);
            float substrateW = vavg(ld3(s->textures[b->albedo].avg));
            float substrateWeight = substrateW*b->avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            float specularProbability = specularWeight/(specularWeight + substrateWeight);
            diffusePdf *= (1.0f - specularProbability);
            glossyPdf *= specularProbability;
        }
        return glossyPdf + diffusePdf;
    }
    case TGHIP_BSDF_MIXED: { /* MixedBsdf.cpp:124-130 */
        float ratio;
        if (!mixed_adjustedRatio(s, b, e, &ratio)) return 0.0f;
        return bsdf_pdf(s, b->sub0, e)*ratio + bsdf_pdf(s, b->sub1, e)*(1.0f - ratio);
    }
    case TGHIP_BSDF_DIFFUSE_TRANSMISSION: { /* DiffuseTransmissionBsdf.cpp:77-88 */
        int sampleR = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0, sampleT = (e->requested & TGHIP_LOBE_DIFFUSE_T) != 0;
        if (!sampleR && !sampleT) return 0.0f;
        float transmittanceProbability = sampleR && sampleT ? b->eta[0] : (sampleR ? 0.0f : 1.0f);
        float factor = e->wi.z*e->wo.z < 0.0f ? transmittanceProbability : 1.0f - transmittanceProbability;
        return factor*cosineHemispherePdf(e->wo);
    }
    case TGHIP_BSDF_PHONG: { /* PhongBsdf.cpp:101-124 */
        int evalGlossy = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0, evalDiffuse = (e->requested & TGHIP_LOBE_DIFFUSE_R) != 0;
        if (!evalGlossy && !evalDiffuse) return 0.0f;
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        float result = 0.0f;
        if (evalGlossy) {
            float cosTheta = vdot(V(-e->wi.x, -e->wi.y, e->wi.z), e->wo);
            if (cosTheta > 0.0f)
                result += powf(cosTheta, b->eta[0])*b->k[1];
        }
        if (evalDiffuse && evalGlossy)
            result = result*(1.0f - b->eta[1]) + b->eta[1]*cosineHemispherePdf(e->wo);
        else if (evalDiffuse)
            result = cosineHemispherePdf(e->wo);
        return result;
    }
    case TGHIP_BSDF_THINSHEET: /* ThinSheetBsdf.cpp:112-119 */
        return ((e->requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e->wi, e->wo)) ? 1.0f : 0.0f;
    case TGHIP_BSDF_OREN_NAYAR: { /* OrenNayarBsdf.cpp:125-135 */
        if (!(e->requested & TGHIP_LOBE_DIFFUSE_R)) return 0.0f;
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        float ratio = fminf(fmaxf(bsdf_roughness(s, b, e), 0.01f), 1.0f);
        return O_INV_TWO_PI*ratio + cosineHemispherePdf(e->wo)*(1.0f - ratio);
    }
    case TGHIP_BSDF_ROUGH_COAT: { /* RoughCoatBsdf.cpp:259-298 */
        int sampleR = (e->requested & TGHIP_LOBE_GLOSSY_R) != 0;
        int sampleT = (e->requested & s->bsdfs[b->sub0].lobes) != 0;
        if (!sampleT && !sampleR) return 0.0f;
        if (e->wi.z <= 0.0f || e->wo.z <= 0.0f) return 0.0f;
        v3 wi = e->wi, wo = e->wo;
        float eta = 1.0f/b->ior;
        float cosThetaTi, cosThetaTo;
        float Fi = dielectricReflectanceT(eta, wi.z, &cosThetaTi);
        float Fo = dielectricReflectanceT(eta, wo.z, &cosThetaTo);
        float specularProbability;
        if (sampleR && sampleT) {
            float substrateWeight = b->avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            specularProbability = specularWeight/(specularWeight + substrateWeight);
        } else {
            specularProbability = sampleR ? 1.0f : 0.0f;
        }
        float glossyPdf = 0.0f;
        if (sampleR)
            glossyPdf = rd_pdfBase(e, 1, 0, bsdf_roughness(s, b, e), b->ior, b->distribution);
        float substratePdf = 0.0f;
        if (sampleT) {
            if (Fi < 1.0f && Fo < 1.0f) {
                Event q = *e;
                q.wi = V(wi.x*eta, wi.y*eta, copysignf(cosThetaTi, wi.z));
                q.wo = V(wo.x*eta, wo.y*eta, copysignf(cosThetaTo, wo.z));
                substratePdf = bsdf_pdf(s, b->sub0, &q);
                substratePdf *= eta*eta*fabsf(wo.z/cosThetaTo);
            }
        }
        return glossyPdf*specularProbability + substratePdf*(1.0f - specularProbability);
    }
    case TGHIP_BSDF_TRANSPARENCY:
        return bsdf_pdf(s, b->sub0, e);
    }
    return 0.0f;
}

/* Bsdf::eta (Bsdf.hpp:99-103; DielectricBsdf.cpp:166-174, RoughDielectricBsdf.cpp:274-280) */
static float bsdf_eta(const TgHipSceneDesc *s, int bi, const Event *e)
{
    const TgHipBsdf *b = &s->bsdfs[bi];
    if (b->type == TGHIP_BSDF_DIELECTRIC || b->type == TGHIP_BSDF_ROUGH_DIELECTRIC) {
        if (e->wi.z*e->wo.z >= 0.0f) return 1.0f;
        return e->wi.z < 0.0f ? b->ior : 1.0f/b->ior;
    }
    return 1.0f;
}
/* radiance-transport wrappers (adjoint == false): Bsdf.hpp:71-97 */
static v3 bsdf_eval_rt(const TgHipSceneDesc *s, int bi, const Event *e) { return vscale(bsdf_eval(s, bi, e), sqr(bsdf_eta(s, bi, e))); }
static int bsdf_sample_rt(const TgHipSceneDesc *s, int bi, Event *e)
{
    if (!bsdf_sample(s, bi, e)) return 0;
    e->weight = vscale(e->weight, sqr(bsdf_eta(s, bi, e)));
    return 1;
}

/* ---------------------------------------------------------------------------------------------
 * Geometry: rays, BVH2 traversal, primitive tests
 * ------------------------------------------------------------------------------------------- */
typedef struct { v3 o, d; float tmin, tmax; } Ray;
typedef struct { uint64_t nodes, prims, rays; } TravStats;

/* Quad::intersect (Quad.cpp:71-98) on a record; returns 1 and shrinks *tmax on a hit */
static int quad_test(const TgHipPrimRec *r, const TgHipObject *o, const Ray *ray, float tmax, float *t, float *l0, float *l1, int *backSide)
{
    v3 n = ld3(o->normal);
    float nDotW = vdot(ray->d, n);
    if (fabsf(nDotW) < 1e-6f)
        return 0;
    v3 base = ld3(r->a);
    float tt = vdot(n, vsub(base, ray->o))/nDotW;
    if (tt < ray->tmin || tt > tmax)
        return 0;
    v3 q = vadd(ray->o, vscale(ray->d, tt));
    v3 v = vsub(q, base);
    float a = vdot(v, ld3(r->b))*r->p0;
    float b = vdot(v, ld3(r->c))*r->p1;
    if (a < 0.0f || a > 1.0f || b < 0.0f || b > 1.0f)
        return 0;
    *t = tt; *l0 = a; *l1 = b; *backSide = nDotW >= 0.0f;
    return 1;
}

/* Cube::intersect (Cube.cpp:94-125) */
static int cube_test(const TgHipObject *o, const Ray *ray, float tmax, float *t, int *backSide)
{
    v3 p = mat3_tmul(o->rot, vsub(ray->o, ld3(o->pos)));
    v3 d = mat3_tmul(o->rot, ray->d);
    float pa[3] = {p.x, p.y, p.z}, da[3] = {d.x, d.y, d.z};
    float ttMin = ray->tmin, ttMax = tmax;
    for (int i = 0; i < 3; ++i) {
        float invD = 1.0f/da[i];
        float relMin = -o->scale[i] - pa[i];
        float relMax = o->scale[i] - pa[i];
        if (invD >= 0.0f) {
            ttMin = fmaxf(ttMin, relMin*invD);
            ttMax = fminf(ttMax, relMax*invD);
        } else {
            ttMax = fminf(ttMax, relMin*invD);
            ttMin = fmaxf(ttMin, relMax*invD);
        }
    }
    if (ttMin <= ttMax) {
        if (ttMin > ray->tmin && ttMin < tmax) { *t = ttMin; *backSide = 0; return 1; }
        else if (ttMax > ray->tmin && ttMax < tmax) { *t = ttMax; *backSide = 1; return 1; }
    }
    return 0;
}

/* Sphere::intersect (Sphere.cpp:69-94); radius in scale[0] */
static int sphere_test(const TgHipObject *o, const Ray *ray, float tmax, float *t, int *backSide)
{
    v3 p = vsub(ray->o, ld3(o->pos));
    float B = vdot(p, ray->d);
    float C = vlensq(p) - o->scale[0]*o->scale[0];
    float detSq = B*B - C;
    if (detSq >= 0.0f) {
        float det = sqrtf(detSq);
        float tt = -B - det;
        if (tt < tmax && tt > ray->tmin) { *t = tt; *backSide = 0; return 1; }
        tt = -B + det;
        if (tt < tmax && tt > ray->tmin) { *t = tt; *backSide = 1; return 1; }
    }
    return 0;
}

/* uv and normal of a point on a cube / sphere (Cube.cpp:157-170, Sphere.cpp:120-129) */
/* Disk::intersect (Disk.cpp:63-85); u = rSq for intersectionInfo, back = -nDotW < _cosApex */
static int disk_test(const TgHipObject *o, const Ray *ray, float tmax, float *t, float *rSq, int *backSide)
{
    v3 n = ld3(o->normal), center = ld3(o->pos);
    float nDotW = vdot(ray->d, n);
    float tt = vdot(n, vsub(center, ray->o))/nDotW;
    if (tt < ray->tmin || tt > tmax)
        return 0;
    v3 q = vadd(ray->o, vscale(ray->d, tt));
    v3 v = vsub(q, center);
    float r2 = vlensq(v);
    if (r2 > o->scale[0]*o->scale[0])
        return 0;
    *t = tt; *rSq = r2; *backSide = -nDotW < o->scale[1];
    return 1;
}
/* Disk::intersectionInfo (Disk.cpp:114-129); hp = the stored hit point ray.pos + t*ray.dir */
static void disk_surface(const TgHipObject *o, v3 hp, float rSq, float *u, float *v)
{
    v3 d = vsub(hp, ld3(o->pos));
    float x = vdot(d, ld3(o->edge1)), y = vdot(d, ld3(o->edge0));      /* bitangent, tangent */
    *v = sqrtf(rSq)/o->scale[0];
    *u = (x == 0.0f && y == 0.0f) ? 0.0f : (atan2f(y, x)*O_INV_TWO_PI + 0.5f);
}

/* Cylinder::intersect (Cylinder.cpp:55-108): pos = _pos, rot = _rot, scale = {_radius, _halfHeight, _capped}.  cap = +-1 when a
 * cap was hit (its sign), 0 for the side. */
static int cylinder_test(const TgHipObject *o, const Ray *ray, float tmax, float *tOut, int *backSide, float *cap)
{
    const float radius = o->scale[0], halfHeight = o->scale[1], invRadius = 1.0f/radius;
    v3 pLocal = mat3_tmul(o->rot, vsub(ray->o, ld3(o->pos)));
    v3 dLocal = mat3_tmul(o->rot, ray->d);
    float px = pLocal.x*invRadius, pz = pLocal.z*invRadius, dx = dLocal.x*invRadius, dz = dLocal.z*invRadius;
    int didHit = 0;
    float farT = tmax;
    if (o->scale[2] != 0.0f && fabsf(dLocal.y) > 1e-6f) {
        for (int k = 0; k < 2; ++k) {
            float sign = k == 0 ? 1.0f : -1.0f;
            float t = (sign*halfHeight - pLocal.y)/dLocal.y;
            if (t > ray->tmin && t < farT) {
                float hx = px + t*dx, hz = pz + t*dz;
                if (hx*hx + hz*hz < 1.0f) {
                    didHit = 1; *cap = sign; *backSide = sign*dLocal.y > 0.0f; farT = t;
                }
            }
        }
    }
    float A = dx*dx + dz*dz, B = px*dx + pz*dz, C = px*px + pz*pz - 1.0f;
    float detSq = B*B - A*C;
    if (detSq >= 0.0f) {
        float det = sqrtf(detSq);
        for (int k = 0; k < 2; ++k) {
            float sign = k == 0 ? 1.0f : -1.0f;
            float t = (-B - sign*det)/A;
            if (t > ray->tmin && t < farT) {
                float h = pLocal.y + dLocal.y*t;
                if (h >= -halfHeight && h <= halfHeight) {
                    didHit = 1; *cap = 0.0f; *backSide = sign < 0.0f; farT = t;
                }
            }
        }
    }
    if (didHit) *tOut = farT;
    return didHit;
}
/* Cylinder::intersectionInfo (Cylinder.cpp:122-132).  The reference keeps what Cylinder::intersect computed in the cylinder's own space --
 * pHit = p + t d in the unit-radius cross-section, h = pLocal.y + dLocal.y t (:70-74, :92-97) -- so the normal and the uv are functions of
 * the RAY and t, not of the world-space hit point (going through it and back costs an ulp in one sample out of ten that meet a cylinder). */
static void cylinder_surface(const TgHipObject *o, const Ray *ray, float t, float cap, v3 *n, float *u, float *v)
{
    const float invRadius = 1.0f/o->scale[0];
    v3 pLocal = mat3_tmul(o->rot, vsub(ray->o, ld3(o->pos)));
    v3 dLocal = mat3_tmul(o->rot, ray->d);
    float px = pLocal.x*invRadius, pz = pLocal.z*invRadius, dx = dLocal.x*invRadius, dz = dLocal.z*invRadius;
    float hx = px + t*dx, hz = pz + t*dz;
    if (cap != 0.0f) {
        *n = mat3_mul(o->rot, V(0.0f, cap, 0.0f));
        *u = hx*0.5f + 0.5f; *v = hz*0.5f + 0.5f;
    } else {
        float h = pLocal.y + dLocal.y*t;
        *n = mat3_mul(o->rot, V(hx, 0.0f, hz));
        *u = atan2f(hz, hx)*O_INV_TWO_PI + 0.5f;
        *v = h*(0.5f/o->scale[1]) + 0.5f;
    }
}

static void cube_surface(const TgHipObject *o, v3 hp, v3 *n, float *u, float *v)
{
    v3 p = mat3_tmul(o->rot, vsub(hp, ld3(o->pos)));
    float pa[3] = {p.x, p.y, p.z};
    float ex[3] = {fabsf(p.x) - o->scale[0], fabsf(p.y) - o->scale[1], fabsf(p.z) - o->scale[2]};
    int dim = ex[0] > ex[1] ? (ex[0] > ex[2] ? 0 : 2) : (ex[1] > ex[2] ? 1 : 2);   /* Vec::maxDim */
    float nn[3] = {0.0f, 0.0f, 0.0f};
    nn[dim] = pa[dim] < 0.0f ? -1.0f : 1.0f;
    float uvw[3];
    for (int i = 0; i < 3; ++i) uvw[i] = (pa[i]/o->scale[i])*0.5f + 0.5f;
    *n = mat3_mul(o->rot, V(nn[0], nn[1], nn[2]));
    *u = uvw[(dim + 1) % 3]; *v = uvw[(dim + 2) % 3];
}
static void sphere_surface(const TgHipObject *o, v3 hp, v3 *n, float *u, float *v)
{
    *n = vdivs(vsub(hp, ld3(o->pos)), o->scale[0]);
    v3 localN = mat3_tmul(o->rot, *n);
    *u = atan2f(localN.y, localN.x)*O_INV_TWO_PI + 0.5f;
    *v = acosf(fclamp(localN.z, -1.0f, 1.0f))*O_INV_PI;
    if (isnan(*u)) *u = 0.0f;
}

/* ---- Embree's arithmetic in its triangle test, restated bit for bit (round 4) ---------------------------------------------------
 * Two details of Embree 2.x decide the last bit of every triangle hit's t / u / v, and with them -- a path being a chaotic function of
 * its hits -- 0.1 % (materialtest) to 1.4 % (a dielectric hero material) of the samples of every scene with a triangle mesh:
 *  (1) its dot product associates from the right: dot(a, b) = a.x*b.x + (a.y*b.y + a.z*b.z) (common/math/vec3.h:182, madd(a.x, b.x,
 *      madd(a.y, b.y, a.z*b.z)); without FMA in the SSE4.2 build the reference's CMake default makes: simd/vfloat4_sse2.h:263);
 *  (2) it does not divide: t = T*rcp(absDen) with rcp(a) = r*(2 - r*a), r = _mm_rcp_ps(a) (kernels/geometry/
 *      triangle_intersector_moeller.h:43-49, simd/vfloat4_sse2.h:166-173) -- the hardware's reciprocal ESTIMATE plus one Newton step.
 * The estimate is the instruction's, not IEEE's: on Intel CPUs (the goldens under tests/golden/ were rendered on one) RCPPS looks the top
 * 11 mantissa bits i up in a table whose entries are 2^25/(4097 + 2 i) rounded to the nearest integer -- a 13-bit significand --, takes
 * the exponent from the operand and flushes denormal operands to zero (-> infinity) and denormal results to zero.  intel_rcpps below is
 * that rule in integer arithmetic; tools/rcpps_sweep.c holds it to the instruction for every one of the 2^32 bit patterns (zero
 * mismatches on the Xeon this repository's goldens come from; AMD's RCPPS is a different function, which is why the rule is restated
 * instead of executed).  With (1) and (2) the oracle's radiance is the reference's BIT FOR BIT in every sample of every mesh case
 * (tests/test_oracle_golden.py: BIT_IDENTICAL), where exact division and a left-to-right dot product left 0.1 - 1.4 % of them on other paths. */
static inline float edot(v3 a, v3 b) { return a.x*b.x + (a.y*b.y + a.z*b.z); }
static inline float intel_rcpps(float x)
{
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u, e = (u >> 23) & 0xffu, i = (u >> 12) & 0x7ffu;
    uint32_t bits;
    if (e == 0u) bits = sign | 0x7f800000u;                               /* zeros and denormals: +-infinity */
    else if (e == 255u) bits = (u & 0x7fffffu) ? (u | 0x00400000u) : sign;   /* NaN quieted; 1/infinity = 0 */
    else if (e >= 253u) bits = sign;                                      /* the result would be denormal: flushed to zero */
    else {
        /* q = 2^25 / d rounded to nearest (d is odd: no ties): the float quotient's floor, corrected by the exact integer remainder */
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
static inline float embree_rcp(float a)
{
    const float r = intel_rcpps(a);
    return r*(2.0f - r*a);                       /* _mm_mul_ps(r, _mm_sub_ps(2, _mm_mul_ps(r, a))): three roundings (-ffp-contract=off) */
}
/* (for the tests: tests/test_host.py holds intel_rcpps to the instruction where the host has Intel's, tests/test_gpu_libm.py holds the
 * device's restatement to this one) */
void oracle_embree_rcp(int raw_estimate, const float *x, float *y, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        y[i] = raw_estimate ? intel_rcpps(x[i]) : embree_rcp(x[i]);
}

/* Embree MoellerTrumboreIntersector1 (thirdparty/embree/kernels/geometry/triangle_intersector_moeller.h:76-113, finalize() :43-49),
 * operation for operation.  Embree's e1 = v0 - v1 = -rec.b, e2 = v2 - v0 = rec.c. */
static int tri_test(const TgHipPrimRec *r, const Ray *ray, float tmax, float *t, float *u, float *v)
{
    v3 v0 = ld3(r->a);
    v3 e1 = vneg(ld3(r->b)), e2 = ld3(r->c);
    v3 Ng = vcross(e1, e2);
    v3 C = vsub(v0, ray->o);
    v3 R = vcross(ray->d, C);
    float den = edot(Ng, ray->d);
    float absDen = fabsf(den);
    float sgn = den < 0.0f ? -1.0f : 1.0f;     /* xor with the sign mask */
    float U = edot(R, e2)*sgn;
    float Vv = edot(R, e1)*sgn;
    if (!(den != 0.0f && U >= 0.0f && Vv >= 0.0f && U + Vv <= absDen))
        return 0;
    float T = edot(Ng, C)*sgn;
    if (!(T > absDen*ray->tmin && T < absDen*tmax))
        return 0;
    const float rcpAbsDen = embree_rcp(absDen);
    *t = T*rcpAbsDen; *u = U*rcpAbsDen; *v = Vv*rcpAbsDen;
    return 1;
}

static inline int box_test(const float *lo, const float *hi, const Ray *ray, v3 invD, float tmax, float *tEntry)
{
    float t0x = (lo[0] - ray->o.x)*invD.x, t1x = (hi[0] - ray->o.x)*invD.x;
    float t0y = (lo[1] - ray->o.y)*invD.y, t1y = (hi[1] - ray->o.y)*invD.y;
    float t0z = (lo[2] - ray->o.z)*invD.z, t1z = (hi[2] - ray->o.z)*invD.z;
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), ray->tmin));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    tf *= 1.0000004f;     /* conservative: never cull a primitive the exact test would accept */
    *tEntry = tn;
    return tn <= tf;
}

/* TraceableScene::intersect (renderer/TraceableScene.hpp:170-192): closest hit over all finite
 * primitives.  The reference does it with Embree BVH4s; we walk the flattened BVH2, near child first. */
/* closest hit so far: TgHipHit plus the instance record the hit triangle was reached through (-1: none) */
typedef struct { float t, u, v; int32_t rec; int32_t inst; } Hit;

/* Quaternion<float>::operator*(Vec3) (math/Quaternion.hpp:78-88); q = (w, x, y, z) */
static v3 quat_rotate(const float *q, v3 o)
{
    float tx = 2.0f*(q[2]*o.z - q[3]*o.y);
    float ty = 2.0f*(q[3]*o.x - q[1]*o.z);
    float tz = 2.0f*(q[1]*o.y - q[2]*o.x);
    return V(o.x + q[0]*tx + q[2]*tz - q[3]*ty,
             o.y + q[0]*ty + q[3]*tx - q[1]*tz,
             o.z + q[0]*tz + q[1]*ty - q[2]*tx);
}
static void instance_quat(const TgHipPrimRec *r, float *q) { q[0] = r->p0; q[1] = r->b[0]; q[2] = r->b[1]; q[3] = r->b[2]; }
static void instance_inv_quat(const TgHipPrimRec *r, float *q) { q[0] = r->p0; q[1] = -r->b[0]; q[2] = -r->b[1]; q[3] = -r->b[2]; }   /* conjugate() :42-45 */

static void bvh_walk(const TgHipSceneDesc *s, int32_t root, const Ray *ray, float *tmax, Hit *hit, TravStats *st, int objFilter, int32_t inst);
static void wide_walk_from(const TgHipSceneDesc *s, int32_t startNode, int32_t startInst, const Ray *worldRay, float *tmax, Hit *hit, TravStats *st, int objFilter);
/* A master's subtree is a plain nearest-hit query (Instance.cpp:296-303 fixes the order BETWEEN instances only; inside, the reference asks the master
 * mesh's own Embree scene).  The device's render kernels walk it through the master's 8-wide subtree (round 6, k_trace_closest_instw), its stand-alone
 * ray query (tghip_trace_rays) through the BVH2; the oracle follows: renders the wide subtree when the scene carries one, oracle_trace_rays the BVH2.
 * Where two triangles of a master answer a ray one rounding apart the first one tested keeps the hit, so the order is part of the answer. */
static int g_inst_wide = 1;

/* ---- Instance::intersect as the reference computes it (primitives/Instance.cpp:290-311) -------------------------------------------
 * The `instances` primitive walks ITS OWN bvh over the instances (bvh/BinaryBvh.hpp:197-287, restated below statement for statement;
 * the tree itself is restated by the host, csrc/host/RefInstanceBvh.cpp, and arrives as BVH2 nodes with the reference's exact child
 * boxes behind the instance-set record).  Every instance whose leaf the walk reaches gets the ray in its master's space with
 * nearT = the distance at which the ray enters that leaf's box and farT = INFINITY (Ray::scatter's default), and a hit there replaces
 * the hit so far -- nearer or not: `ray.setFarT(localRay.farT())`.  What bounds the damage is the walk's own tMax, the MINIMUM of the
 * hit distances so far (:274), which culls the boxes that begin behind it.  So the result is the LAST hit in the tree's visiting
 * order among the instances whose boxes begin in front of every hit found before them -- a function of the reference's tree, its
 * near/far rule (`minMax[0] < minMax[1]`: the RIGHT child first on a tie) and its pop test, all of which are kept here.
 * (Rounds 1-3 returned the nearest hit instead; 2.3 % of the samples of the crowded golden scene saw another surface.) */
static inline float ref_max(float a, float b) { return a > b ? a : b; }     /* _mm_max_ps / Tungsten::max: the second operand when unordered */
static inline float ref_min(float a, float b) { return a < b ? a : b; }
/* BinaryBvh::bboxIntersection (:155-178) */
static int ref_bbox_intersection(const float *lo, const float *hi, const Ray *ray, float *tMin, float *tMax)
{
    const float o[3] = {ray->o.x, ray->o.y, ray->o.z}, d[3] = {ray->d.x, ray->d.y, ray->d.z};
    float ttMin = *tMin, ttMax = *tMax;
    for (int i = 0; i < 3; ++i) {
        const float invD = 1.0f/d[i], relMin = lo[i] - o[i], relMax = hi[i] - o[i];
        if (invD >= 0.0f) {
            ttMin = ref_max(ttMin, relMin*invD);
            ttMax = ref_min(ttMax, relMax*invD);
        } else {
            ttMax = ref_min(ttMax, relMin*invD);
            ttMin = ref_max(ttMin, relMax*invD);
        }
    }
    if (ttMin <= ttMax) { *tMin = ttMin; *tMax = ttMax; return 1; }
    return 0;
}
/* one child of a node in BinaryBvh::trace (:231-242): entry distance and -- negated, as the reference carries it -- exit distance, clipped
 * to [nearT, farT]; near / far planes by the sign of the direction (>= 0: keep), the products with 1/d and -1/d */
static inline int ref_child_test(const float *lo, const float *hi, const float *o, const float *d, const float *invD, float nearT, float farT, float *tEntry)
{
    float tn[3], ntf[3];
    for (int k = 0; k < 3; ++k) {
        const float nearP = d[k] >= 0.0f ? lo[k] : hi[k], farP = d[k] >= 0.0f ? hi[k] : lo[k];
        tn[k] = (nearP - o[k])*invD[k];
        ntf[k] = (farP - o[k])*(-invD[k]);
    }
    const float tmin = ref_max(ref_max(ref_max(tn[0], tn[1]), tn[2]), nearT);
    const float ntmax = ref_max(ref_max(ref_max(ntf[0], ntf[1]), ntf[2]), -farT);
    *tEntry = tmin;
    return tmin <= -ntmax;
}
static void instance_set_walk(const TgHipSceneDesc *s, uint32_t setRec, const Ray *ray, float *rayFarT, Hit *hit, TravStats *st)
{
    const TgHipPrimRec *set = &s->recs[setRec];
    struct { int32_t node; float tMin; } stack[TGHIP_MAX_BVH_DEPTH + 2];
    int sp = 0;
    float tMin = ray->tmin, tMax = *rayFarT;
    if (!ref_bbox_intersection(set->a, set->b, ray, &tMin, &tMax))
        return;
    const float o[3] = {ray->o.x, ray->o.y, ray->o.z}, d[3] = {ray->d.x, ray->d.y, ray->d.z};
    const float invD[3] = {1.0f/d[0], 1.0f/d[1], 1.0f/d[2]};
    const float nearT = ray->tmin;
    float farT = *rayFarT;                        /* nearFar[2..3], negated: the ray's farT until the first leaf, the walk's tMax after it */
    int32_t node;
    memcpy(&node, &set->c[0], 4);
    for (;;) {
        while (node >= 0) {
            const TgHipBvhNode *n = &s->nodes[node];
            if (st) st->nodes++;
            float e0, e1;
            const int hitL = ref_child_test(n->lo0, n->hi0, o, d, invD, nearT, farT, &e0);
            const int hitR = ref_child_test(n->lo1, n->hi1, o, d, invD, nearT, farT, &e1);
            if (hitL && hitR) {
                if (e0 < e1) { stack[sp].node = n->child1; stack[sp].tMin = e1; ++sp; node = n->child0; tMin = e0; }
                else         { stack[sp].node = n->child0; stack[sp].tMin = e0; ++sp; node = n->child1; tMin = e1; }
            } else if (hitL) { node = n->child0; tMin = e0; }
            else if (hitR) { node = n->child1; tMin = e1; }
            else goto pop;
        }
        {
            const uint32_t first = TGHIP_LEAF_FIRST(node), count = TGHIP_LEAF_COUNT(node);
            for (uint32_t k = first; k < first + count; ++k) {
                /* the intersector lambda of Instance::intersect (:294-303) */
                const uint32_t ri = s->inst_prims[k];
                const TgHipPrimRec *r = &s->recs[ri];
                if (st) st->prims++;
                float q[4];
                instance_inv_quat(r, q);
                Ray local = {quat_rotate(q, vsub(ray->o, ld3(r->a))), quat_rotate(q, ray->d), tMin, INFINITY};
                uint32_t root;
                memcpy(&root, &r->c[0], 4);
                float localFarT = INFINITY;
                Hit lh;
                lh.rec = -1; lh.inst = -1; lh.t = localFarT; lh.u = lh.v = 0.0f;
                uint32_t wideRoot;
                memcpy(&wideRoot, &r->c[2], 4);
                if (g_inst_wide && s->wide_nodes && s->num_wide_nodes && wideRoot != 0u)
                    wide_walk_from(s, (int32_t)wideRoot, (int32_t)ri, &local, &localFarT, &lh, st, -1);
                else
                    bvh_walk(s, (int32_t)root, &local, &localFarT, &lh, st, -1, (int32_t)ri);
                if (lh.rec >= 0) { *hit = lh; *rayFarT = localFarT; }
            }
            tMax = ref_min(tMax, *rayFarT);
            farT = tMax;
        }
pop:
        for (;;) {
            if (sp == 0) return;
            --sp;
            node = stack[sp].node;
            tMin = stack[sp].tMin;
            if (!(tMax < tMin)) break;
        }
    }
}

static void test_rec(const TgHipSceneDesc *s, uint32_t i, const Ray *ray, float *tmax, Hit *hit, TravStats *st, int objFilter, int32_t inst)
{
    const TgHipPrimRec *r = &s->recs[i];
    if (st) st->prims++;
    float t, u = 0.0f, v = 0.0f; int back = 0;
    int ok = 0;
    switch (TGHIP_REC_KIND(r->meta)) {
    case TGHIP_REC_TRIANGLE: ok = tri_test(r, ray, *tmax, &t, &u, &v); break;
    case TGHIP_REC_QUAD: ok = quad_test(r, &s->objects[TGHIP_REC_OBJECT(r->meta)], ray, *tmax, &t, &u, &v, &back); break;
    case TGHIP_REC_CUBE: ok = cube_test(&s->objects[TGHIP_REC_OBJECT(r->meta)], ray, *tmax, &t, &back); u = (float)back; break;
    case TGHIP_REC_SPHERE: ok = sphere_test(&s->objects[TGHIP_REC_OBJECT(r->meta)], ray, *tmax, &t, &back); u = (float)back; break;
    case TGHIP_REC_DISK: ok = disk_test(&s->objects[TGHIP_REC_OBJECT(r->meta)], ray, *tmax, &t, &v, &back); u = (float)back; break;   /* v carries rSq */
    case TGHIP_REC_CYLINDER: ok = cylinder_test(&s->objects[TGHIP_REC_OBJECT(r->meta)], ray, *tmax, &t, &back, &v); u = (float)back; break;   /* v carries the cap sign */
    case TGHIP_REC_INSTANCE_SET:
        instance_set_walk(s, i, ray, tmax, hit, st);
        return;
    default: break;
    }
    (void)objFilter;
    if (ok) { *tmax = t; hit->t = t; hit->u = u; hit->v = v; hit->rec = (int32_t)i; hit->inst = inst; }
}

static void bvh_walk(const TgHipSceneDesc *s, int32_t root, const Ray *ray, float *tmax, Hit *hit, TravStats *st, int objFilter, int32_t inst)
{
    int32_t stack[TGHIP_MAX_BVH_DEPTH + 2];
    int sp = 0;
    v3 invD = V(1.0f/ray->d.x, 1.0f/ray->d.y, 1.0f/ray->d.z);
    int32_t cur = root;
    for (;;) {
        if (cur >= 0) {
            const TgHipBvhNode *n = &s->nodes[cur];
            if (st) st->nodes++;
            float e0, e1;
            int h0 = box_test(n->lo0, n->hi0, ray, invD, *tmax, &e0);
            int h1 = box_test(n->lo1, n->hi1, ray, invD, *tmax, &e1);
            if (h0 && h1) {
                if (e1 < e0) { stack[sp++] = n->child0; cur = n->child1; }
                else { stack[sp++] = n->child1; cur = n->child0; }
                continue;
            } else if (h0) { cur = n->child0; continue; }
            else if (h1) { cur = n->child1; continue; }
        } else {
            uint32_t first = TGHIP_LEAF_FIRST(cur), count = TGHIP_LEAF_COUNT(cur);
            for (uint32_t i = first; i < first + count; ++i)
                if (objFilter < 0 || (int)TGHIP_REC_OBJECT(s->recs[i].meta) == objFilter)
                    test_rec(s, i, ray, tmax, hit, st, objFilter, inst);
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
}

/* ---- the 8-wide BVH with quantised child boxes (include/tungsten_hip.h: TgHipWideNode) ------------------------------
 * The same per-ray machine as the device's (tungsten_amd/csrc/hip/pt_kernels.h: wideNext / wideVisit), statement for
 * statement: the records of a node's hit leaf children are tested first (ascending record order), then its hit internal
 * children are visited in ascending (slot XOR ray octant) order, the ones not taken yet waiting on a stack of groups;
 * plane distances are fmaf(q, spacing/d, (origin - o)/d) -- one correctly rounded operation on both sides -- so node and
 * record visit counts agree exactly (tests/test_gpu_parity.py::test_trace_rays_matches_oracle_exactly). */
static int g_use_wide = 0;      /* oracle_set_wide_bvh: closest-hit queries of oracle_trace_rays walk the wide BVH */
static inline uint32_t wide_permute(uint32_t h, uint32_t oct)
{
    if (oct & 1u) h = ((h & 0x55u) << 1) | ((h >> 1) & 0x55u);
    if (oct & 2u) h = ((h & 0x33u) << 2) | ((h >> 2) & 0x33u);
    if (oct & 4u) h = ((h & 0x0Fu) << 4) | ((h >> 4) & 0x0Fu);
    return h;
}
static inline float wide_spacing(uint8_t e) { union { uint32_t u; float f; } c; c.u = (uint32_t)e << 23; return c.f; }
static inline float wide_inv(float d) { return 1.0f/(fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d); }
#define WIDE_LEAVE     0xFFFFFFFFu
#define WIDE_RECS_FLAG 0x80000000u
static void wide_walk_from(const TgHipSceneDesc *s, int32_t startNode, int32_t startInst, const Ray *worldRay, float *tmax, Hit *hit, TravStats *st, int objFilter)
{
    struct { uint32_t base, masks; } stack[TGHIP_MAX_WIDE_DEPTH + 2];
    int sp = 0;
    Ray ray = *worldRay;
    float idir[3], org[3];
    uint32_t octInv;
#define WIDE_RAY_SETUP() do { idir[0] = wide_inv(ray.d.x); idir[1] = wide_inv(ray.d.y); idir[2] = wide_inv(ray.d.z); \
        org[0] = ray.o.x; org[1] = ray.o.y; org[2] = ray.o.z; \
        octInv = (idir[0] < 0.0f ? 1u : 0u) | (idir[1] < 0.0f ? 2u : 0u) | (idir[2] < 0.0f ? 4u : 0u); } while (0)
    WIDE_RAY_SETUP();
    uint32_t grpBase = 0, grpMasks = 0, triBase = 0, triMask = 0, triValid = 0, curNode = 0;
    int32_t node = startNode, curInst = startInst;
    for (;;) {
        if (triMask) {
            uint32_t b = (uint32_t)__builtin_ctz(triMask);
            uint32_t i = triBase + (uint32_t)__builtin_popcount(triValid & ((1u << b) - 1u));
            triMask &= triMask - 1u;
            const TgHipPrimRec *r = &s->recs[i];
            if (TGHIP_REC_KIND(r->meta) == TGHIP_REC_INSTANCE) {
                /* wideEnterInstance: what is left of this node waits on the stack, the ray goes into the master's space */
                if (st) st->prims++;
                if (grpMasks & 0xFFu) { stack[sp].base = grpBase; stack[sp].masks = grpMasks; ++sp; }
                if (triMask) { stack[sp].base = WIDE_RECS_FLAG | curNode; stack[sp].masks = triMask; ++sp; }
                stack[sp].base = WIDE_LEAVE; stack[sp].masks = 0; ++sp;
                grpMasks = 0; triMask = 0;
                float q[4];
                instance_inv_quat(r, q);
                ray.o = quat_rotate(q, vsub(ray.o, ld3(r->a)));
                ray.d = quat_rotate(q, ray.d);
                WIDE_RAY_SETUP();
                uint32_t root;
                memcpy(&root, &r->c[2], 4);
                node = (int32_t)root;
                curInst = (int32_t)i;
                continue;
            }
            if (objFilter < 0 || (int)TGHIP_REC_OBJECT(r->meta) == objFilter)
                test_rec(s, i, &ray, tmax, hit, st, objFilter, curInst);
            else if (st) st->prims++;            /* (the device fetches the record either way) */
            continue;
        }
        if (node < 0) {
            if ((grpMasks & 0xFFu) == 0u) {
                if (sp == 0) break;
                --sp;
                if (stack[sp].base == WIDE_LEAVE) {              /* back to world space */
                    ray.o = worldRay->o; ray.d = worldRay->d;
                    WIDE_RAY_SETUP();
                    curInst = -1;
                    continue;
                }
                if (stack[sp].base & WIDE_RECS_FLAG) {           /* the rest of a top-level node's records */
                    curNode = stack[sp].base & ~WIDE_RECS_FLAG;
                    triMask = stack[sp].masks;
                    triBase = s->wide_nodes[curNode].rec_base;
                    triValid = s->wide_nodes[curNode].leaf_valid;
                    continue;
                }
                grpBase = stack[sp].base; grpMasks = stack[sp].masks;
            }
            uint32_t hits = grpMasks & 0xFFu, imask = grpMasks >> 8;
            uint32_t slot = (uint32_t)__builtin_ctz(hits) ^ octInv;
            node = (int32_t)(grpBase + (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u)));
            hits &= hits - 1u;
            grpMasks = (imask << 8) | hits;
            if (hits) { stack[sp].base = grpBase; stack[sp].masks = grpMasks; ++sp; }
        }
        const TgHipWideNode *n = &s->wide_nodes[node];
        curNode = (uint32_t)node;
        node = -1;
        if (st) st->nodes++;
        float adjS[3], adjO[3];
        for (int a = 0; a < 3; ++a) {
            adjS[a] = wide_spacing(n->exp[a])*idir[a];
            adjO[a] = (n->origin[a] - org[a])*idir[a];
        }
        uint32_t hitmask = 0;
        for (int sl = 0; sl < 8; ++sl) {
            float tn = ray.tmin, tf = *tmax;
            float tnA[3], tfA[3];
            for (int a = 0; a < 3; ++a) {
                const int neg = (octInv >> a) & 1u;
                const float qn = (float)(neg ? n->qhi[a][sl] : n->qlo[a][sl]), qf = (float)(neg ? n->qlo[a][sl] : n->qhi[a][sl]);
                tnA[a] = fmaf(qn, adjS[a], adjO[a]);
                tfA[a] = fmaf(qf, adjS[a], adjO[a]);
            }
            tn = fmaxf(fmaxf(tnA[0], tnA[1]), fmaxf(tnA[2], tn));
            tf = fminf(fminf(tfA[0], tfA[1]), fminf(tfA[2], tf));
            tf *= 1.0000004f;
            if (tn <= tf)
                hitmask |= 1u << sl;
        }
        grpBase = n->child_base;
        grpMasks = ((uint32_t)n->imask << 8) | wide_permute(hitmask & n->imask, octInv);
        triBase = n->rec_base;
        triValid = n->leaf_valid;
        triMask = 0;
        for (int sl = 0; sl < 8; ++sl)
            if ((hitmask & ~(uint32_t)n->imask) >> sl & 1u) triMask |= 15u << (4*sl);
        triMask &= triValid;
    }
#undef WIDE_RAY_SETUP
}
static void wide_walk(const TgHipSceneDesc *s, const Ray *worldRay, float *tmax, Hit *hit, TravStats *st, int objFilter)
{
    wide_walk_from(s, 0, -1, worldRay, tmax, hit, st, objFilter);
}
void oracle_set_wide_bvh(int on) { g_use_wide = on; }
void oracle_set_inst_wide(int on) { g_inst_wide = on; }     /* 0: masters of instanced scenes through their BVH2 (the device's option inst_wide = 0) */

/* objFilter >= 0: only records of that object are tested (a mesh light's own rtcIntersect, TriangleMesh.cpp:317-335) */
/* A flat list of analytic primitives (quads, cubes, spheres, disks, cylinders) is intersected by WALKING THE REFERENCE'S TREE: TraceableScene::intersect -> rtcIntersect over Embree's
 * BVH4 with one primitive per leaf (TgHipSceneDesc::top_nodes, include/tungsten_hip.h: TgHipTopNode -- built by csrc/host/EmbreeTopTree.cpp,
 * which tests/test_top_tree.py holds to trees read out of the reference's own Embree), visited as BVH4Intersector1 visits it
 * (thirdparty/embree/kernels/bvh/bvh_intersector1.cpp:60-125, bvh_traverser1.h:41-104, common/stack_item.h:39-60):
 *   - a child counts only when the ray passes its box by the node's slab test (embree_box_near: embree_box_visible's arithmetic) under the hit
 *     distance found so far (`ray_far = ray.tfar` after every leaf);
 *   - one child hit: descend; two: the first only `if (d0 < d1)`, the other is pushed; three / four: all pushed, sorted by Embree's
 *     networks on the entry distances' bit patterns, the nearest popped;
 *   - a node or leaf popped behind the hit so far is skipped (`stackPtr->dist > ray.tfar`) -- by its BOX's entry distance, which can lie an
 *     ulp behind the distance the primitive's own intersect() would report.
 * With coincident faces (the Cornell box's blocks stand ON the floor; a light lies IN the ceiling; a Sobol' point on the image's diagonal sends
 * its ray into the seam of floor and wall) these rules, not the distance alone, decide which primitive a ray hits: 99 samples in 12 golden cases
 * and the seam samples of the stress renders followed another path before.
 * Scenes without top_nodes (a triangle mesh is ONE item of the reference's tree with an Embree scene of its own; here its triangles live in ONE
 * tree with the other records) keep the plain list / the BVH. */
static int embree_box_near(const Ray *ray, v3 lo, v3 hi, float *tNearOut)
{
    const float o[3] = {ray->o.x, ray->o.y, ray->o.z}, d[3] = {ray->d.x, ray->d.y, ray->d.z};
    const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    float tNear = fmaxf(ray->tmin, 0.0f), tFar = fmaxf(ray->tmax, 0.0f);
    float n[3], f[3];
    for (int k = 0; k < 3; ++k) {
        float a = fabsf(d[k]) < 1e-18f ? 1e-18f : d[k];           /* zero_fix */
        float rdir = embree_rcp(a);
        n[k] = ((rdir >= 0.0f ? l[k] : h[k]) - o[k])*rdir;
        f[k] = ((rdir >= 0.0f ? h[k] : l[k]) - o[k])*rdir;
    }
    int32_t in[3], ifr[3], it, ift;                               /* maxi / mini: on the bit patterns as signed integers */
    memcpy(in, n, 12); memcpy(ifr, f, 12); memcpy(&it, &tNear, 4); memcpy(&ift, &tFar, 4);
    int32_t nxy = in[0] > in[1] ? in[0] : in[1], nzt = in[2] > it ? in[2] : it, nn = nxy > nzt ? nxy : nzt;
    int32_t fxy = ifr[0] > ifr[1] ? ifr[1] : ifr[0], fzt = ifr[2] > ift ? ift : ifr[2], ff = fxy > fzt ? fzt : fxy;
    memcpy(tNearOut, &nn, 4);
    return !(nn > ff);
}
/* Quad::bounds (Quad.cpp:281-289), Cube::bounds (Cube.cpp:333-344), Sphere::bounds (Sphere.cpp:273-276), Disk::bounds, Cylinder::bounds from the
 * flattened object; 0 for a record kind whose bounds are not restated (triangles).  (csrc/host/EmbreeTopTree.cpp: referenceLeafBounds is the library's copy.) */
int oracle_leaf_bounds(const TgHipSceneDesc *s, uint32_t i, float lo3[3], float hi3[3])
{
    const TgHipPrimRec *r = &s->recs[i];
    const TgHipObject *o = &s->objects[TGHIP_REC_OBJECT(r->meta)];
    v3 p[8]; int n = 0;
    switch (TGHIP_REC_KIND(r->meta)) {
    case TGHIP_REC_QUAD: { v3 b = ld3(o->base), e0 = ld3(o->edge0), e1 = ld3(o->edge1); p[0] = b; p[1] = vadd(b, e0); p[2] = vadd(b, e1); p[3] = vadd(vadd(b, e0), e1); n = 4; break; }
    case TGHIP_REC_CUBE:
        for (int k = 0; k < 8; ++k)
            p[k] = vadd(ld3(o->pos), mat3_mul(o->rot, V((k & 1) ? o->scale[0] : -o->scale[0], (k & 2) ? o->scale[1] : -o->scale[1], (k & 4) ? o->scale[2] : -o->scale[2])));
        n = 8; break;
    case TGHIP_REC_SPHERE: { v3 c = ld3(o->pos); float rr = o->scale[0]; p[0] = V(c.x - rr, c.y - rr, c.z - rr); p[1] = V(c.x + rr, c.y + rr, c.z + rr); n = 2; break; }
    case TGHIP_REC_DISK: {          /* Disk::bounds (Disk.cpp:298-306): _center -+ _frame.tangent*_r -+ _frame.bitangent*_r */
        v3 c = ld3(o->pos), t = vscale(ld3(o->edge0), o->scale[0]), b = vscale(ld3(o->edge1), o->scale[0]);
        p[0] = vsub(vsub(c, t), b); p[1] = vsub(vadd(c, t), b); p[2] = vadd(vadd(c, t), b); p[3] = vadd(vsub(c, t), b); n = 4; break; }
    case TGHIP_REC_CYLINDER: {      /* Cylinder::bounds (Cylinder.cpp:272-279): _pos +- _axis*_halfHeight, then grow(_radius); _axis = the second column of _rot (Mat4f.cpp:40-47) */
        v3 c = ld3(o->pos), a = vscale(V(o->rot[1], o->rot[4], o->rot[7]), o->scale[1]);
        p[0] = vadd(c, a); p[1] = vsub(c, a); n = 2; break; }
    default: return 0;
    }
    v3 lo = p[0], hi = p[0];
    for (int k = 1; k < n; ++k) {
        lo = V(p[k].x < lo.x ? p[k].x : lo.x, p[k].y < lo.y ? p[k].y : lo.y, p[k].z < lo.z ? p[k].z : lo.z);   /* Box::grow: min(_min, p), MathUtil.hpp:33-52 */
        hi = V(p[k].x > hi.x ? p[k].x : hi.x, p[k].y > hi.y ? p[k].y : hi.y, p[k].z > hi.z ? p[k].z : hi.z);
    }
    if (TGHIP_REC_KIND(r->meta) == TGHIP_REC_CYLINDER) { float rr = o->scale[0]; lo = V(lo.x - rr, lo.y - rr, lo.z - rr); hi = V(hi.x + rr, hi.y + rr, hi.z + rr); }
    lo3[0] = lo.x; lo3[1] = lo.y; lo3[2] = lo.z; hi3[0] = hi.x; hi3[1] = hi.y; hi3[2] = hi.z;
    return 1;
}
static int g_flat_order = 1;
void oracle_set_flat_order(int on) { g_flat_order = on; }     /* 0: the plain list in record order (tests: what the order changes) */
/* PROTOTYPE for the next round (tools/top_tree_meshes.py; OFF in every test, the device has no counterpart): scenes WITH triangle meshes walked
 * as the reference walks them -- the top-level tree over ALL finite primitives, a mesh being one item whose leaf runs the mesh's own closest-hit
 * query (TriangleMesh::intersect, TriangleMesh.cpp:317-335) under the hit distance so far.  With it on, a leaf of top_nodes names an OBJECT
 * (~object index) instead of a record.  The mesh's own Embree BVH4 is not restated: inside a mesh this library's BVH2 decides ties. */
static int g_top_items = 0;
void oracle_set_top_items(int on) { g_top_items = on; }
typedef struct { int32_t ref; uint32_t dist; } TopStackItem;
static void top_swap(TopStackItem *a, TopStackItem *b) { TopStackItem t = *a; *a = *b; *b = t; }
static int embree_top_walk(const TgHipSceneDesc *s, const Ray *ray0, Hit *hit, TravStats *st, int objFilter)
{
    Ray ray = *ray0;                                                /* ray.tmax is Embree's ray.tfar */
    float rayFar = fmaxf(ray.tmax, 0.0f);                           /* ray_far: max(ray.tfar, 0) at first, ray.tfar after a leaf */
    TopStackItem stack[64];
    int sp = 0;
    stack[sp].ref = 0; stack[sp].dist = 0xFF800000u; sp++;           /* root, dist = neg_inf */
    if (st) st->prims += s->num_recs;                                /* the unit both sides count: the list's records, once per ray */
    while (sp > 0) {
        --sp;
        int32_t cur = stack[sp].ref;
        float popDist; memcpy(&popDist, &stack[sp].dist, 4);
        if (popDist > ray.tmax) continue;                            /* popped behind the hit so far */
        int descend = 1;
        while (cur >= 0) {
            const TgHipTopNode *n = &s->top_nodes[cur];
            int hitSlot[4], nh = 0; uint32_t hitDist[4];
            Ray probe = ray; probe.tmax = rayFar;
            for (int i = 0; i < 4; ++i) {
                if (n->child[i] == TGHIP_TOP_EMPTY) continue;
                float tn;
                if (!embree_box_near(&probe, ld3(n->lower[i]), ld3(n->upper[i]), &tn)) continue;
                hitSlot[nh] = i; memcpy(&hitDist[nh], &tn, 4); nh++;
            }
            if (nh == 0) { descend = 0; break; }
            if (nh == 1) { cur = n->child[hitSlot[0]]; continue; }
            if (nh == 2) {
                int32_t c0 = n->child[hitSlot[0]], c1 = n->child[hitSlot[1]];
                if (hitDist[0] < hitDist[1]) { stack[sp].ref = c1; stack[sp].dist = hitDist[1]; sp++; cur = c0; }
                else                         { stack[sp].ref = c0; stack[sp].dist = hitDist[0]; sp++; cur = c1; }
                continue;
            }
            for (int k = 0; k < nh; ++k) { stack[sp].ref = n->child[hitSlot[k]]; stack[sp].dist = hitDist[k]; sp++; }
            TopStackItem *s1 = &stack[sp - 1], *s2 = &stack[sp - 2], *s3 = &stack[sp - 3];
            if (nh == 3) {
                if (s2->dist < s1->dist) top_swap(s2, s1);
                if (s3->dist < s2->dist) top_swap(s3, s2);
                if (s2->dist < s1->dist) top_swap(s2, s1);
            } else {
                TopStackItem *s4 = &stack[sp - 4];
                if (s2->dist < s1->dist) top_swap(s2, s1);
                if (s4->dist < s3->dist) top_swap(s4, s3);
                if (s3->dist < s1->dist) top_swap(s3, s1);
                if (s4->dist < s2->dist) top_swap(s4, s2);
                if (s3->dist < s2->dist) top_swap(s3, s2);
            }
            cur = stack[sp - 1].ref; sp--;
        }
        if (!descend) continue;
        if (g_top_items) {                                           /* prototype: the leaf is an object */
            const int obj = (int)~cur;
            if (objFilter < 0 || obj == objFilter) {
                if (s->objects[obj].type == TGHIP_OBJ_MESH) {
                    bvh_walk(s, 0, &ray, &ray.tmax, hit, NULL, obj, -1);
                } else {
                    for (uint32_t r = 0; r < s->num_recs; ++r)          /* (flat scan: the prototype's scenes are small) */
                        if ((int)TGHIP_REC_OBJECT(s->recs[r].meta) == obj && TGHIP_REC_KIND(s->recs[r].meta) != TGHIP_REC_TRIANGLE &&
                            TGHIP_REC_KIND(s->recs[r].meta) != TGHIP_REC_INSTANCE) {       /* (an `instances` primitive: its set record) */
                            test_rec(s, r, &ray, &ray.tmax, hit, NULL, objFilter, -1);
                            break;
                        }
                }
            }
            rayFar = ray.tmax;
            continue;
        }
        const uint32_t rec = (uint32_t)~cur;
        if (objFilter < 0 || (int)TGHIP_REC_OBJECT(s->recs[rec].meta) == objFilter)
            test_rec(s, rec, &ray, &ray.tmax, hit, NULL, objFilter, -1);
        rayFar = ray.tmax;
    }
    return hit->rec >= 0;
}

/* how many of a render's rays the device's shortcut (oracle_flat_device_form below) decides without the walk: tools only */
static int g_flat_stats = 0;
static uint64_t g_flat_rays = 0, g_flat_walked = 0;
static int flat_shortcut_decides(const TgHipSceneDesc *s, const Ray *ray, Hit *got);
static void flat_stats_count(const TgHipSceneDesc *s, const Ray *ray)
{
    Hit h;
    const int dec = flat_shortcut_decides(s, ray, &h);
#pragma omp atomic
    g_flat_rays++;
    if (!dec) {
#pragma omp atomic
        g_flat_walked++;
    }
}
void oracle_flat_stats(int enable, uint64_t *rays, uint64_t *walked)
{
    if (rays) *rays = g_flat_rays;
    if (walked) *walked = g_flat_walked;
    g_flat_stats = enable; g_flat_rays = 0; g_flat_walked = 0;
}
static int scene_intersect_obj(const TgHipSceneDesc *s, const Ray *ray, Hit *hit, TravStats *st, int objFilter)
{
    float tmax = ray->tmax;
    hit->rec = -1; hit->inst = -1; hit->t = tmax; hit->u = hit->v = 0.0f;
    if (st) st->rays++;
    if (s->top_nodes && s->num_top_nodes && g_flat_order) {
        if (g_flat_stats) flat_stats_count(s, ray);
        return embree_top_walk(s, ray, hit, st, objFilter);
    }
    if (s->num_recs <= TGHIP_FLAT_MAX_RECS && s->num_instances == 0) {           /* flat list (include/tungsten_hip.h) */
        for (uint32_t i = 0; i < s->num_recs; ++i)
            if (objFilter < 0 || (int)TGHIP_REC_OBJECT(s->recs[i].meta) == objFilter)
                test_rec(s, i, ray, &tmax, hit, st, objFilter, -1);
        return hit->rec >= 0;
    }
    if (g_use_wide && s->wide_nodes && s->num_instances == 0)   /* (with instances the wide BVH serves any-hit queries only: closest hits are a matter of the reference's visiting order) */
        wide_walk(s, ray, &tmax, hit, st, objFilter);
    else
        bvh_walk(s, 0, ray, &tmax, hit, st, objFilter, -1);
    return hit->rec >= 0;
}
static int scene_intersect(const TgHipSceneDesc *s, const Ray *ray, Hit *hit, TravStats *st)
{
    return scene_intersect_obj(s, ray, hit, st, -1);
}

/* IntersectionInfo (primitives/IntersectionInfo.hpp:11-22) + what hitBackside() needs */
typedef struct {
    v3 Ng, Ns, p, w;
    float u, v, epsilon;
    int object, bsdf, backSide;
    /* Primitive::tangentSpace of the primitive that was hit (TriangleMesh.cpp:362-384, Quad.cpp:133-139, Cube.cpp:172-182, Sphere.cpp:131-137,
     * Disk.cpp:129-140, Cylinder.cpp:135-141; Instance.cpp:348-351 has none): read by the bump-mapped shading frame only */
    v3 T, B;
    int hasTB;
} Info;

static void intersection_info(const TgHipSceneDesc *s, const Ray *ray, const Hit *hit, Info *info)
{
    const TgHipPrimRec *r = &s->recs[hit->rec];
    int objIdx = (int)TGHIP_REC_OBJECT(r->meta);
    const TgHipObject *o = &s->objects[objIdx];
    info->object = objIdx;
    info->p = vadd(ray->o, vscale(ray->d, hit->t));      /* TraceableScene.hpp:184 */
    info->w = ray->d;
    info->epsilon = 5e-4f;                               /* DefaultEpsilon, TraceableScene.hpp:39 */
    info->T = info->B = V(0.0f, 0.0f, 0.0f);
    info->hasTB = 0;
    switch (TGHIP_REC_KIND(r->meta)) {
    case TGHIP_REC_TRIANGLE: {   /* TriangleMesh.cpp:317-355, 80-106 */
        const TgHipTriAttr *a = &s->tri_attrs[hit->rec];
        {   /* TriangleMesh::tangentSpace (:362-384); the record holds p0, p1 - p0, p2 - p0 */
            v3 q1 = ld3(r->b), q2 = ld3(r->c);
            float s1 = a->uv1[0] - a->uv0[0], t1 = a->uv1[1] - a->uv0[1];
            float s2 = a->uv2[0] - a->uv0[0], t2 = a->uv2[1] - a->uv0[1];
            float invDet = s1*t2 - s2*t1;
            if (!(fabsf(invDet) < 1e-6f) && hit->inst < 0) {
                info->T = vnorm(vsub(vscale(q1, t2), vscale(q2, t1)));
                info->B = vnorm(vsub(vscale(q2, s1), vscale(q1, s2)));
                info->hasTB = 1;
            }
        }
        v3 NgU = vcross(ld3(r->b), ld3(r->c));
        v3 dLocal = ray->d;
        if (hit->inst >= 0) {       /* the master was intersected with the ray in its own space (Instance.cpp:295-297) */
            float qi[4];
            instance_inv_quat(&s->recs[hit->inst], qi);
            dLocal = quat_rotate(qi, ray->d);
        }
        info->backSide = vdot(NgU, dLocal) > 0.0f;
        info->Ng = vnorm(NgU);
        float u = hit->u, v = hit->v;
        if (o->flags & TGHIP_OBJF_SMOOTH) {
            v3 n = vadd(vadd(vscale(ld3(a->n0), 1.0f - u - v), vscale(ld3(a->n1), u)), vscale(ld3(a->n2), v));
            info->Ns = vnorm(n);
        } else {
            info->Ns = info->Ng;
        }
        info->u = (1.0f - u - v)*a->uv0[0] + u*a->uv1[0] + v*a->uv2[0];
        info->v = (1.0f - u - v)*a->uv0[1] + u*a->uv1[1] + v*a->uv2[1];
        info->bsdf = a->bsdf;
        break;
    }
    case TGHIP_REC_QUAD:         /* Quad.cpp:123-131 */
        info->Ng = info->Ns = ld3(o->normal);
        info->u = hit->u; info->v = hit->v;
        info->bsdf = o->bsdf;
        info->backSide = vdot(ray->d, ld3(o->normal)) >= 0.0f;
        info->T = ld3(o->edge0); info->B = ld3(o->edge1); info->hasTB = 1;   /* Quad.cpp:133-139 */
        break;
    case TGHIP_REC_CUBE:         /* Cube.cpp:157-170 */
        cube_surface(o, info->p, &info->Ng, &info->u, &info->v);
        info->Ns = info->Ng;
        info->bsdf = o->bsdf;
        info->backSide = hit->u != 0.0f;
        {   /* Cube::tangentSpace (:172-182) */
            v3 lp = mat3_tmul(o->rot, vsub(info->p, ld3(o->pos)));
            float ex[3] = {fabsf(lp.x) - o->scale[0], fabsf(lp.y) - o->scale[1], fabsf(lp.z) - o->scale[2]};
            int dim = 0;
            if (ex[1] > ex[dim]) dim = 1;
            if (ex[2] > ex[dim]) dim = 2;
            float t[3] = {0.0f, 0.0f, 0.0f}, b[3] = {0.0f, 0.0f, 0.0f};
            t[(dim + 1) % 3] = 1.0f; b[(dim + 2) % 3] = 1.0f;
            info->T = mat3_mul(o->rot, V(t[0], t[1], t[2]));
            info->B = mat3_mul(o->rot, V(b[0], b[1], b[2]));
            info->hasTB = 1;
        }
        break;
    case TGHIP_REC_SPHERE:       /* Sphere.cpp:120-129 */
        sphere_surface(o, info->p, &info->Ng, &info->u, &info->v);
        info->Ns = info->Ng;
        info->bsdf = o->bsdf;
        info->backSide = hit->u != 0.0f;
        {   /* Sphere::tangentSpace (:131-137) */
            v3 localN = mat3_tmul(o->rot, info->Ng);
            info->T = mat3_mul(o->rot, V(-localN.y, localN.x, localN.z));
            info->B = vcross(info->Ns, info->T);
            info->hasTB = 1;
        }
        break;
    case TGHIP_REC_CYLINDER:     /* Cylinder.cpp:122-132 */
        cylinder_surface(o, ray, hit->t, hit->v, &info->Ng, &info->u, &info->v);
        info->Ns = info->Ng;
        info->bsdf = o->bsdf;
        info->backSide = hit->u != 0.0f;
        info->T = ld3(o->normal); info->B = vcross(info->Ng, info->T); info->hasTB = 1;   /* Cylinder.cpp:135-141: T = _axis */
        break;
    case TGHIP_REC_DISK:         /* Disk.cpp:114-129 */
        info->Ng = info->Ns = ld3(o->normal);
        disk_surface(o, info->p, hit->v, &info->u, &info->v);
        info->bsdf = o->bsdf;
        info->backSide = hit->u != 0.0f;
        {   /* Disk::tangentSpace (:129-140) */
            v3 dd = vsub(info->p, ld3(o->pos));
            if (vlensq(dd) != 0.0f) {
                dd = vnorm(dd);
                info->T = vcross(ld3(o->normal), dd);
                info->B = dd;
                info->hasTB = 1;
            }
        }
        break;
    default:
        info->Ng = info->Ns = V(0, 1, 0); info->u = info->v = 0; info->bsdf = o->bsdf; info->backSide = 0;
        break;
    }
    if (hit->inst >= 0) {
        /* Instance::intersectionInfo (primitives/Instance.cpp:337-346): the master's normals go to world space; info.p,
         * which TraceableScene::intersect already set to the WORLD-space hit point (TraceableScene.hpp:184), is transformed
         * once more -- the reference's behaviour, reproduced as it is (NEE and MIS are evaluated from that point) */
        const TgHipPrimRec *ir = &s->recs[hit->inst];
        float q[4];
        instance_quat(ir, q);
        info->Ng = quat_rotate(q, info->Ng);
        info->Ns = quat_rotate(q, info->Ns);
        info->p = vadd(ld3(ir->a), quat_rotate(q, info->p));
        info->object = (int)TGHIP_REC_OBJECT(ir->meta);
        info->hasTB = 0;                                 /* Instance::tangentSpace (Instance.cpp:348-351) */
    }
}

/* InfiniteSphere::directionToUV (InfiniteSphere.cpp:27-39) */
static void inf_directionToUV(const TgHipObject *o, v3 wi, float *u, float *v, float *sinTheta)
{
    /* Skydome::directionToUV (Skydome.cpp:41-50) is the same map without the primitive's rotation */
    v3 wLocal = (o->flags & TGHIP_OBJF_SKYDOME) ? wi : mat3_tmul(o->rot, wi);
    if (sinTheta) *sinTheta = sqrtf(fmaxf(1.0f - wLocal.y*wLocal.y, 0.0f));
    *u = atan2f(wLocal.z, wLocal.x)*O_INV_TWO_PI + 0.5f;
    *v = acosf(-wLocal.y)*O_INV_PI;
}
static v3 inf_uvToDirection(const TgHipObject *o, float u, float v, float *sinTheta)   /* :41-51 */
{
    float phi = (u - 0.5f)*O_TWO_PI;
    float theta = v*O_PI;
    *sinTheta = sinf(theta);
    v3 wLocal = V(cosf(phi)**sinTheta, -cosf(theta), sinf(phi)**sinTheta);
    return (o->flags & TGHIP_OBJF_SKYDOME) ? wLocal : mat3_mul(o->rot, wLocal);          /* Skydome.cpp:51-61 */
}

/* ---------------------------------------------------------------------------------------------
 * The integrator: TraceBase + PathTracer
 * ------------------------------------------------------------------------------------------- */
/* what one sample adds to the output buffers (PathTracer.cpp:78-96, 133-140): v = depth | normal | albedo | visibility
 * in TgHipAuxPixel channel order (3..10), has[output] = whether the sample recorded it */
typedef struct AuxSample { int has[TGHIP_AUX_OUTPUTS]; float v[TGHIP_AUX_CHANNELS]; } AuxSample;

typedef struct {
    const TgHipSceneDesc *s;
    Sampler *sampler;
    TravStats *st;
    uint64_t shadow_rays, closest_rays;
    struct AuxSample *aux;          /* auxiliary output values of the sample in flight (NULL: _trackOutputValues off) */
    v3 *visRequest;                 /* handleSurface's `transmittance` out-parameter (TraceBase.cpp:520): where lightSample's shadow result goes */
    v3 *transmittanceOut;           /* = visRequest while lightSample's attenuatedEmission runs (:275), else NULL (bsdfSample / volume: nullptr) */
} Ctx;

typedef struct { Frame frame; v3 wi; int flipped; } Local;   /* the part of SurfaceScatterEvent that persists */

/* BitmapTexture::derivatives (textures/BitmapTexture.cpp:359-398) of a scalar bitmap: central differences of the four texels around the
 * lookup, interpolated; constant and checker textures have none (ConstantTexture.cpp:60-63, CheckerTexture.cpp:76-79) */
static void texture_derivatives(const TgHipSceneDesc *s, int texIdx, float u0, float v0, float *du, float *dv)
{
    const TgHipTexture *t = &s->textures[texIdx];
    *du = *dv = 0.0f;
    if (t->type != TGHIP_TEX_BITMAP)
        return;
    const int w = t->w, h = t->h;
    const float *tex = s->texels + t->texel_offset;
    float u = u0*w - 0.5f;
    float v = (1.0f - v0)*h - 0.5f;
    int iu = (int)u, iv = (int)v;
    u -= iu; v -= iv;
    iu = ((iu % w) + w) % w;
    iv = ((iv % h) + h) % h;
    int x0 = iu - 1, x1 = iu, x2 = (iu + 1) % w, x3 = (iu + 2) % w;
    int y0 = iv - 1, y1 = iv, y2 = (iv + 1) % h, y3 = (iv + 2) % h;
    if (x0 < 0) x0 = w - 1;
    if (y0 < 0) y0 = h - 1;
#define TEXEL(x, y) ((t->flags & TGHIP_TEXF_RGB) ? (tex[((size_t)(x) + (size_t)(y)*w)*3] + tex[((size_t)(x) + (size_t)(y)*w)*3 + 1] + tex[((size_t)(x) + (size_t)(y)*w)*3 + 2])/3.0f : tex[(size_t)(x) + (size_t)(y)*w])
    float a01 = TEXEL(x1, y0), a02 = TEXEL(x2, y0);
    float a10 = TEXEL(x0, y1), a11 = TEXEL(x1, y1), a12 = TEXEL(x2, y1), a13 = TEXEL(x3, y1);
    float a20 = TEXEL(x0, y2), a21 = TEXEL(x1, y2), a22 = TEXEL(x2, y2), a23 = TEXEL(x3, y2);
    float a31 = TEXEL(x1, y3), a32 = TEXEL(x2, y3);
#undef TEXEL
    float du11 = a12 - a10, du12 = a13 - a11, du21 = a22 - a20, du22 = a23 - a21;
    float dv11 = a21 - a01, dv21 = a31 - a11, dv12 = a22 - a02, dv22 = a32 - a12;
    *du = ((du11*(1.0f - u) + du12*u)*(1.0f - v) + (du21*(1.0f - u) + du22*u)*v)*t->scale;
    *dv = ((dv11*(1.0f - u) + dv12*u)*(1.0f - v) + (dv21*(1.0f - u) + dv22*u)*v)*t->scale;
}

/* Primitive::setupTangentFrame (primitives/Primitive.cpp:125-163): the frame of the shading normal -- unless the bsdf carries a
 * non-constant bump map (TgHipBsdf::bump1): then tangent and bitangent come from the primitive's tangent space, tilted by the map's
 * derivatives.  (Anisotropic lobes, the other reason for the long way, belong to the hair bcsdfs only.) */
static Frame shading_frame(const TgHipSceneDesc *s, const Info *info)
{
    const int bump = s->bsdfs[info->bsdf].bump1 - 1;
    if (bump < 0 || !info->hasTB)
        return frame_from_normal(info->Ns);
    v3 T = info->T, B = info->B, N = info->Ns;
    float du, dv;
    texture_derivatives(s, bump, info->u, info->v, &du, &dv);
    T = vadd(T, vscale(info->Ns, du - vdot(info->Ns, T)));
    B = vadd(B, vscale(info->Ns, dv - vdot(info->Ns, B)));
    N = vcross(T, B);
    if (N.x == 0.0f && N.y == 0.0f && N.z == 0.0f)
        return frame_from_normal(info->Ns);
    if (vdot(N, info->Ns) < 0.0f)
        N = vneg(N);
    N = vnorm(N);
    T = vsub(T, vscale(N, vdot(N, T)));
    if (T.x == 0.0f && T.y == 0.0f && T.z == 0.0f)
        return frame_from_normal(info->Ns);
    T = vnorm(T);
    Frame f;
    f.normal = N; f.tangent = T; f.bitangent = vcross(N, T);
    return f;
}

static Local makeLocalScatterEvent(const Ctx *c, const Info *info, const Ray *ray)   /* TraceBase.cpp:24-51 */
{
    Local l;
    l.frame = shading_frame(c->s, info);        /* Primitive::setupTangentFrame (Primitive.cpp:125-163) */
    int hitBackside = vdot(l.frame.normal, ray->d) > 0.0f;
    int isTransmissive = (c->s->bsdfs[info->bsdf].lobes & LOBE_TRANSMISSIVE) != 0;
    l.flipped = c->s->settings.enable_two_sided_shading && hitBackside && !isTransmissive;
    if (l.flipped) {
        l.frame.normal = vneg(l.frame.normal);
        l.frame.tangent = vneg(l.frame.tangent);
    }
    l.wi = toLocal(&l.frame, vneg(ray->d));
    return l;
}

static int isConsistent(const Ctx *c, const Info *info, const Local *l, v3 woLocal, v3 w)   /* TraceBase.cpp:53-60 */
{
    if (!c->s->settings.enable_consistency_checks)
        return 1;
    int geometricBackside = vdot(w, info->Ng) < 0.0f;
    int shadingBackside = (woLocal.z < 0.0f) ^ l->flipped;
    return geometricBackside == shadingBackside;
}

static Event make_event(const Ctx *c, const Info *info, const Local *l, uint32_t requested)
{
    Event e;
    e.wi = l->wi; e.wo = vs(0.0f); e.weight = vs(1.0f); e.pdf = 1.0f;
    e.requested = requested; e.sampled = 0;
    e.u = info->u; e.v = info->v;
    e.sampler = c->sampler;
    return e;
}

/* ---- participating media: HomogeneousMedium (media/HomogeneousMedium.cpp) with the exponential, linear, quadratic,
 * double-exponential, pulse and Erlang transmittances (transmittances/*.cpp).  The reference evaluates the exponential one through
 * fmath's table-based exp (math/FastMath.hpp:14-27), restated below (fmath_exp). ---- */
typedef struct { int firstScatter; int bounce; } MediumState;     /* Medium.hpp:30-47 */
typedef struct { v3 p; float t; v3 weight; int exited; int medium; } MediumSample;
/* FastMath::exp -> fmath::exp / exp_ps (math/FastMath.hpp:14-27; thirdparty/fmath/fmath.hpp:91-126, 221-241, 320-363): the table of
 * 2^(i/1024) built with powf at start-up exactly as fmath's ExpVar constructor builds it, then float and integer arithmetic only */
static uint32_t g_fmath_tbl[1024];
static float g_fmath_a, g_fmath_b;
__attribute__((constructor)) static void fmath_exp_init(void)
{
    float log_2 = logf(2.0f);
    g_fmath_a = 1024/log_2;
    g_fmath_b = log_2/1024;
    for (int i = 0; i < 1024; i++) {
        float y = powf(2.0f, (float)i/1024);
        uint32_t bits;
        memcpy(&bits, &y, 4);
        g_fmath_tbl[i] = bits & 0x7FFFFFu;
    }
}
const uint32_t *oracle_fmath_exp_table(void) { return g_fmath_tbl; }      /* tests/test_media.py: the device carries the same table */
static float fmath_exp(float x)
{
    int32_t xi;
    memcpy(&xi, &x, 4);
    if ((xi & 0x7fffffff) > 0x42b00000)
        x = fmaxf(fminf(x, 88.0f), -88.0f);
    int32_t r = (int32_t)lrintf(x*g_fmath_a);                    /* cvtss2si / cvtps2dq: round to nearest even */
    float t = x - (float)r*g_fmath_b;
    uint32_t bits = ((uint32_t)((r >> 10) + 127) << 23) | g_fmath_tbl[r & 1023];
    float f;
    memcpy(&f, &bits, 4);
    return (1.0f + t)*f;
}
float oracle_fmath_exp(float x) { return fmath_exp(x); }
static inline v3 vexpneg(v3 tau) { return V(fmath_exp(-tau.x), fmath_exp(-tau.y), fmath_exp(-tau.z)); }

/* Primitive::selectMedium (Primitive.hpp:177-183) */
static int selectMedium(const TgHipObject *o, int current, int geometricBackside)
{
    if (o->int_medium >= 0 || o->ext_medium >= 0)
        return geometricBackside ? o->int_medium : o->ext_medium;
    return current;
}

/* ---- transmittances (transmittances/*.cpp): the four kernels surfaceSurface / surfaceMedium / mediumSurface / mediumMedium
 * of one channel, sigmaBar and the two distance samplers.  k: 0 = SS, 1 = SM, 2 = MS, 3 = MM. ---- */
static float trans_leaf_kernel(const TgHipMedium *m, int k, float tau)
{
    const float *p = m->trans_p;
    switch (m->trans_type) {
    case TGHIP_TRANS_LINEAR: {                          /* LinearTransmittance.cpp:32-57 */
        float maxT = p[0];
        if (k == 0) return 1.0f - fminf(tau/maxT, 1.0f);
        if (k == 1) return tau > maxT ? 0.0f : 1.0f/maxT;
        if (k == 2) return tau > maxT ? 0.0f : 1.0f;
        return fabsf(tau - maxT) < 1e-3f ? 1.0f : 0.0f;
    }
    case TGHIP_TRANS_QUADRATIC: {                       /* QuadraticTransmittance.cpp:32-52 */
        float maxT = p[0], t = fminf(tau/maxT, 1.0f);
        if (k == 0) return 1.0f - 2.0f*t + t*t;
        if (k == 1) return (2.0f/maxT)*(1.0f - t);
        if (k == 2) return 1.0f - t;
        return tau > maxT ? 0.0f : 1.0f/maxT;
    }
    case TGHIP_TRANS_DOUBLE_EXPONENTIAL: {              /* DoubleExponentialTransmittance.cpp:34-49 */
        float a = p[0], b = p[1], ea = expf(-a*tau), eb = expf(-b*tau);
        if (k == 0) return 0.5f*(ea + eb);
        if (k == 1) return 0.5f*(a*ea + b*eb);
        if (k == 2) return (a*ea + b*eb)/(a + b);
        return (sqr(a)*ea + sqr(b)*eb)/(a + b);
    }
    case TGHIP_TRANS_PULSE: {                           /* PulseTransmittance.cpp:45-82 */
        float a = p[0], b = p[1], n = p[2];
        int num = (int)n;
        if (k == 0) {
            float idxF = n*(tau - a)/(b - a) + 0.5f;
            idxF = fminf(fmaxf(idxF, 0.0f), n);
            int idx = (int)idxF;
            float height = (float)(num - idx)/n;
            float cellIntegral = height*(idxF - (float)idx);
            if (idx > 0) cellIntegral += ((float)idx - 0.5f) - (float)(idx*(idx - 1))/(2.0f*n);
            else         cellIntegral -= 0.5f;
            return 1.0f - (2.0f/n)*cellIntegral;
        }
        if (k == 1 || k == 2) {
            int idx = (int)(n*(tau - a)/(b - a) + 0.5f);
            idx = idx < 0 ? 0 : (idx > num ? num : idx);
            float ms = 1.0f - (float)idx/n;
            return k == 2 ? ms : 2.0f/(b - a)*ms;
        }
        float idxF = fminf(fmaxf(n*(tau - a)/(b - a), 0.0f), n);
        int idx = (int)idxF;
        return (1.0f/n)*(fabsf(idxF - (float)idx - 0.5f) < 1e-3f ? 1.0f : 0.0f);
    }
    case TGHIP_TRANS_ERLANG: {                          /* ErlangTransmittance.cpp:32-47 */
        float l = p[0], e = expf(-l*tau);
        if (k == 0) return 0.5f*e*(2.0f + l*tau);
        if (k == 1) return e*(1.0f + l*tau)*l*0.5f;
        if (k == 2) return e*(1.0f + l*tau);
        return sqr(l)*tau*e;
    }
    case TGHIP_TRANS_DAVIS: {                           /* DavisTransmittance.cpp:34-49 */
        float alpha = p[0];
        if (k == 0) return powf(1.0f + tau/alpha, -alpha);
        if (k == 1 || k == 2) return powf(1.0f + tau/alpha, -(alpha + 1.0f));
        return (1.0f + 1.0f/alpha)*powf(1.0f + tau/alpha, -(alpha + 2.0f));
    }
    case TGHIP_TRANS_DAVIS_WEINSTEIN: {                 /* DavisWeinsteinTransmittance.cpp:39-82; NaN -> 0 */
        float beta = 2.0f*p[0] - 1.0f;
        float alpha = powf(tau, 1 - beta)/powf(p[1], 1 + beta);             /* computeAlpha */
        float base = 1.0f + tau/alpha;
        float trSurface = powf(base, -alpha), Tr;
        if (k == 0) {
            Tr = trSurface;
        } else if (k == 1 || k == 2) {
            Tr = trSurface*(beta/base - (beta - 1.0f)*alpha/tau*logf(base));
        } else {
            float logBase = logf(base);
            float term1 = beta*(-1.0f + beta*(1.0f + tau) + (-1.0f + 2.0f*beta)*tau/alpha)/(tau*base*base);
            float term2 = ((-1.0f + beta)*beta*alpha/(tau*tau)*(2.0f*tau + base)*logBase)/base;
            float term3 = (beta - 1.0f)*alpha/tau*logBase;
            Tr = trSurface*(term1 - term2 + term3*term3);
        }
        return isnan(Tr) ? 0.0f : Tr;
    }
    default:                                            /* ExponentialTransmittance.cpp:26-41: FastMath::exp */
        return fmath_exp(-tau);
    }
}
static float trans_leaf_sigmaBar(const TgHipMedium *m)
{
    switch (m->trans_type) {
    case TGHIP_TRANS_LINEAR: return 1.0f/m->trans_p[0];
    case TGHIP_TRANS_QUADRATIC: return 2.0f/m->trans_p[0];
    case TGHIP_TRANS_DOUBLE_EXPONENTIAL: return 0.5f*(m->trans_p[0] + m->trans_p[1]);
    case TGHIP_TRANS_PULSE: return 2.0f/(m->trans_p[1] - m->trans_p[0]);
    case TGHIP_TRANS_ERLANG: return m->trans_p[0]*0.5f;
    default: return 1.0f;
    }
}
static v3 trans_leaf_kernel3(const TgHipMedium *m, int k, v3 tau)
{
    if (m->trans_type == TGHIP_TRANS_DAVIS_WEINSTEIN)    /* evaluated on the first channel only (tau[0]) and broadcast (:46-49) */
        return vs(trans_leaf_kernel(m, k, tau.x));
    return V(trans_leaf_kernel(m, k, tau.x), trans_leaf_kernel(m, k, tau.y), trans_leaf_kernel(m, k, tau.z));
}
/* InterpolatedTransmittance (InterpolatedTransmittance.cpp:34-72): operands A = m[1], B = m[2] (include/tungsten_hip.h) */
static inline float lerpf(float a, float b, float u) { return a*(1.0f - u) + b*u; }
static int trans_isDirac(const TgHipMedium *m) { return m->trans_type == TGHIP_TRANS_LINEAR || m->trans_type == TGHIP_TRANS_PULSE; }
static float trans_sigmaBar(const TgHipMedium *m)
{
    if (m->trans_type != TGHIP_TRANS_INTERPOLATED) return trans_leaf_sigmaBar(m);
    return 1.0f/lerpf(1.0f/trans_leaf_sigmaBar(m + 1), 1.0f/trans_leaf_sigmaBar(m + 2), m->trans_p[0]);
}
/* a, b: the operands' kernel k (mediumSurface for k = 1: surfaceMedium = mediumSurface*sigmaBar) */
static float trans_interpolate(const TgHipMedium *m, int k, float a, float b)
{
    const float u = m->trans_p[0];
    if (k == 0) return trans_sigmaBar(m)*lerpf(a/trans_leaf_sigmaBar(m + 1), b/trans_leaf_sigmaBar(m + 2), u);
    if (k == 1) return lerpf(a, b, u)*trans_sigmaBar(m);
    if (k == 2) return lerpf(a, b, u);
    int diracA = trans_isDirac(m + 1) && a > 0.0f, diracB = trans_isDirac(m + 2) && b > 0.0f;
    if (diracA ^ diracB) return diracA ? a : b;
    return lerpf(a, b, u);
}
static float trans_kernel(const TgHipMedium *m, int k, float tau)
{
    if (m->trans_type != TGHIP_TRANS_INTERPOLATED) return trans_leaf_kernel(m, k, tau);
    int kk = k == 1 ? 2 : k;
    return trans_interpolate(m, k, trans_leaf_kernel(m + 1, kk, tau), trans_leaf_kernel(m + 2, kk, tau));
}
static v3 trans_kernel3(const TgHipMedium *m, int k, v3 tau)
{
    if (m->trans_type != TGHIP_TRANS_INTERPOLATED) return trans_leaf_kernel3(m, k, tau);
    int kk = k == 1 ? 2 : k;
    v3 a = trans_leaf_kernel3(m + 1, kk, tau), b = trans_leaf_kernel3(m + 2, kk, tau);
    return V(trans_interpolate(m, k, a.x, b.x), trans_interpolate(m, k, a.y, b.y), trans_interpolate(m, k, a.z, b.z));
}
/* Transmittance::eval / surfaceProbability / mediumPdf (Transmittance.hpp:22-43) */
static v3 trans_eval(const TgHipMedium *m, v3 tau, int startOnSurface, int endOnSurface)
{
    if (startOnSurface && endOnSurface) return trans_kernel3(m, 0, tau);
    if (!startOnSurface && !endOnSurface) return vdivs(trans_kernel3(m, 3, tau), trans_sigmaBar(m));
    return trans_kernel3(m, 2, tau);
}
/* Transmittance::sample = sampleSurface / sampleMedium */
static float trans_leaf_sample(const TgHipMedium *m, Sampler *smp, int startOnSurface)
{
    const float *p = m->trans_p;
    switch (m->trans_type) {
    case TGHIP_TRANS_LINEAR:                            /* :67-74 */
        return startOnSurface ? p[0]*next1D(smp) : p[0];
    case TGHIP_TRANS_QUADRATIC:                         /* :59-66 */
        return startOnSurface ? p[0]*(1.0f - sqrtf(1.0f - next1D(smp))) : p[0]*next1D(smp);
    case TGHIP_TRANS_DOUBLE_EXPONENTIAL: {              /* :56-65 */
        float t = -logf(1.0f - next1D(smp));
        float pa = startOnSurface ? 0.5f : p[0]/(p[0] + p[1]);
        return nextBoolean(smp, pa) ? t/p[0] : t/p[1];
    }
    case TGHIP_TRANS_PULSE: {                           /* :89-108 */
        float a = p[0], b = p[1], n = p[2];
        int num = (int)n;
        if (!startOnSurface)
            return a + (0.5f + (float)(int)(next1D(smp)*n))/n*(b - a);
        float xi = next1D(smp)*n*0.5f;
        float delta = 1.0f/n;
        for (int i = 0; i < num; ++i) {
            float h0 = 1.0f - ((float)i + 0.0f)*delta;
            float h1 = 1.0f - ((float)i + 1.0f)*delta;
            xi -= h0*0.5f;
            if (xi < 0.0f)
                return a + ((float)i + 0.0f + 0.5f*next1D(smp))*(b - a)*delta;
            xi -= h1*0.5f;
            if (xi < 0.0f)
                return a + ((float)i + 0.5f + 0.5f*next1D(smp))*(b - a)*delta;
        }
        return 0.0f;
    }
    case TGHIP_TRANS_ERLANG: {                          /* :54-68 */
        if (!startOnSurface) {
            float x0 = next1D(smp), x1 = next1D(smp);
            return -1.0f/p[0]*logf(x0*x1);
        }
        float xi = next1D(smp);
        float x = 0.5f;
        for (int i = 0; i < 10; ++i) {
            x += (xi - (1.0f - trans_leaf_kernel(m, 0, x)))/trans_leaf_kernel(m, 1, x);
            x = fmaxf(x, 0.0f);
        }
        return x;
    }
    case TGHIP_TRANS_DAVIS:                             /* :56-63 */
        return startOnSurface ? p[0]*(powf(1.0f - next1D(smp), -1.0f/p[0]) - 1.0f)
                              : p[0]*(powf(1.0f - next1D(smp), -1.0f/(1.0f + p[0])) - 1.0f);
    case TGHIP_TRANS_DAVIS_WEINSTEIN: {                 /* :89-118: bisection on the cdf 1 - surfaceSurface / 1 - mediumSurface */
        float xi = next1D(smp);
        float step = 1e6f, result = step*2;
        while (step > 1e-6) {
            float cdf = 1.0f - trans_leaf_kernel(m, startOnSurface ? 0 : 2, result);
            if (cdf > xi) result -= step; else result += step;
            step /= 2;
        }
        return result;
    }
    default:                                            /* ExponentialTransmittance.cpp:46-53 */
        return -logf(1.0f - next1D(smp));
    }
}

static float trans_sample(const TgHipMedium *m, Sampler *smp, int startOnSurface)   /* InterpolatedTransmittance.cpp:65-72 */
{
    if (m->trans_type != TGHIP_TRANS_INTERPOLATED) return trans_leaf_sample(m, smp, startOnSurface);
    return nextBoolean(smp, m->trans_p[0]) ? trans_leaf_sample(m + 2, smp, startOnSurface) : trans_leaf_sample(m + 1, smp, startOnSurface);
}

/* HomogeneousMedium::sampleDistance (HomogeneousMedium.cpp:66-107); state.firstScatter plays "startOnSurface" */
/* ExponentialMedium::densityIntegral / inverseOpticalDepth / density (ExponentialMedium.cpp:76-104) */
static float expmed_densityIntegral(float x, float dx, float tMax)
{
    if (tMax == INFINITY)
        return expf(-x)/dx;
    else if (dx == 0.0f)
        return expf(-x)*tMax;
    else
        return (expf(-x) - expf(-dx*tMax - x))/dx;
}
static float expmed_inverseOpticalDepth(float x, float dx, float tau)
{
    if (dx == 0.0f) {
        return tau/expf(-x);
    } else {
        float denom = 1.0f - dx*expf(x)*tau;
        return denom <= 0.0f ? INFINITY : -logf(denom)/dx;
    }
}
/* ExponentialMedium::sampleDistance (ExponentialMedium.cpp:106-150); exponential transmittance only (include/tungsten_hip.h) */
static int expmed_sampleDistance(const TgHipMedium *m, int medium, Sampler *smp, const Ray *ray, MediumState *state, MediumSample *ms)
{
    if (state->bounce > m->max_bounce)
        return 0;
    const v3 sigmaT = ld3(m->sigma_t);
    float x = m->falloff_scale*vdot(vsub(ray->o, ld3(m->unit_point)), ld3(m->falloff_dir));
    float dx = m->falloff_scale*vdot(ray->d, ld3(m->falloff_dir));
    float maxT = ray->tmax;
    if (m->absorption_only) {
        if (maxT == INFINITY && dx <= 0.0f)
            return 0;
        ms->t = maxT;
        v3 tau = vscale(sigmaT, expmed_densityIntegral(x, dx, ray->tmax));
        ms->weight = trans_eval(m, tau, state->firstScatter, 1);
        ms->exited = 1;
    } else {
        int component = (int)(nextSupplemental(smp)*3);           /* sampler.nextDiscrete(3) */
        float sigmaTc = component == 0 ? sigmaT.x : component == 1 ? sigmaT.y : sigmaT.z;
        float tauC = trans_sample(m, smp, state->firstScatter)/sigmaTc;
        float t = expmed_inverseOpticalDepth(x, dx, tauC);
        ms->t = fminf(t, maxT);
        v3 tau = vscale(sigmaT, expmed_densityIntegral(x, dx, ms->t));
        ms->exited = t >= maxT;
        ms->weight = trans_eval(m, tau, state->firstScatter, ms->exited);      /* (the exponential transmittance does not look at the flags) */
        float pdf;
        if (ms->exited) {
            pdf = vavg(trans_kernel3(m, state->firstScatter ? 0 : 2, tau));
        } else {
            float rho = expf(-(x + dx*ms->t));                      /* density(x, dx, t) */
            pdf = vavg(vmul(vscale(sigmaT, rho), trans_kernel3(m, state->firstScatter ? 1 : 3, tau)));   /* (rho*_sigmaT*mediumPdf).avg() */
            ms->weight = vmul(ms->weight, vscale(vscale(ld3(m->sigma_s), rho), trans_sigmaBar(m)));     /* *= rho*_sigmaS*sigmaBar() */
        }
        ms->weight = vdivs(ms->weight, pdf);
        state->firstScatter = 0; state->bounce++;
    }
    ms->p = vadd(ray->o, vscale(ray->d, ms->t));
    ms->medium = medium;
    return 1;
}

/* ---- AtmosphericMedium (media/AtmosphericMedium.cpp): TgHipMedium carries s = _effectiveFalloffScale in falloff_scale, _center in unit_point and
 * _radius in falloff_dir[0] (include/tungsten_hip.h).  Erf::erfc / erfDifference on floats (math/Erf.hpp:247-283), Erf::erfInv on doubles
 * (math/Erf.hpp:192-245; its tables: erfinv_table.h, generated by tools/gen_erfinv_tables.py), std::erf / std::exp / std::log = this host's libm. */
#include "erfinv_table.h"
#define O_SQRT_PI     1.77245385091f           /* math/Angle.hpp:15-16 */
#define O_INV_SQRT_PI (1.0f/O_SQRT_PI)
static float erf_erfc(float x)
{
    const float p = 0.32759f;
    const float as[5] = {0.254829592f, -0.284496736f, 1.421413741f, -1.453152027f, 1.061405429f};
    float t = 1.0f/(1.0f + p*fabsf(x));
    float ti = copysignf(t*expf(-x*x), x);
    float result = 0.0f;
    for (int i = 0; i < 5; ++i) {
        result += as[i]*ti;
        ti *= t;
    }
    float constant = 1.0f - copysignf(1.0f, x);
    return constant + result;
}
static float erf_erfDifference(float x0, float x1)
{
    const float p = 0.32759f;
    const float as[5] = {0.254829592f, -0.284496736f, 1.421413741f, -1.453152027f, 1.061405429f};
    float t0 = 1.0f/(1.0f + p*fabsf(x0));
    float t1 = 1.0f/(1.0f + p*fabsf(x1));
    float ti0 = copysignf(t0*expf(-x0*x0), x0);
    float ti1 = copysignf(t1*expf(-x1*x1), x1);
    float result = 0.0f;
    for (int i = 0; i < 5; ++i) {
        result += as[i]*(ti0 - ti1);
        ti0 *= t0;
        ti1 *= t1;
    }
    float constant = copysignf(1.0f, x1) - copysignf(1.0f, x0);
    return constant + result;
}
static double poly_eval(int n, double x, const double *P)     /* Polynomial::eval (math/Polynomial.hpp:9-18) */
{
    double result = P[n - 1];
    for (int i = n - 2; i >= 0; --i) {
        result *= x;
        result += P[i];
    }
    return result;
}
static double erf_inv(double z)                                /* Erf::erfInv (math/Erf.hpp:192-245) */
{
    double p, q, sgn;
    if (z < 0) { p = -z; q = 1 - p; sgn = -1; }
    else       { p = z;  q = 1 - z; sgn = 1; }
    double result = 0.0;
    if (p <= 0.5) {
        double g = p*(p + 10.0);
        double r = poly_eval(8, p, ERFINV_P1)/poly_eval(10, p, ERFINV_Q1);
        result = g*ERFINV_Y[0] + g*r;
    } else if (q >= 0.25) {
        double g = sqrt(-2.0*log(q));
        double xs = q - 0.25;
        double r = poly_eval(9, xs, ERFINV_P2)/poly_eval(9, xs, ERFINV_Q2);
        result = g/(ERFINV_Y[1] + r);
    } else {
        double x = sqrt(-log(q));
        if (x < 3.0) {
            double xs = x - 1.125;
            double R = poly_eval(11, xs, ERFINV_P3)/poly_eval(8, xs, ERFINV_Q3);
            result = ERFINV_Y[2]*x + R*x;
        } else if (x < 6.0) {
            double xs = x - 3;
            double R = poly_eval(9, xs, ERFINV_P4)/poly_eval(7, xs, ERFINV_Q4);
            result = ERFINV_Y[3]*x + R*x;
        } else if (x < 18.0) {
            double xs = x - 6.0;
            double R = poly_eval(9, xs, ERFINV_P5)/poly_eval(7, xs, ERFINV_Q5);
            result = ERFINV_Y[4]*x + R*x;
        } else if (x < 44.0) {
            double xs = x - 18.0;
            double R = poly_eval(8, xs, ERFINV_P6)/poly_eval(7, xs, ERFINV_Q6);
            result = ERFINV_Y[5]*x + R*x;
        } else {
            double xs = x - 44.0;
            double R = poly_eval(8, xs, ERFINV_P7)/poly_eval(7, xs, ERFINV_Q7);
            result = ERFINV_Y[6]*x + R*x;
        }
    }
    return sgn*result;
}
/* AtmosphericMedium::density(h, t0) / densityIntegral / inverseOpticalDepth (AtmosphericMedium.cpp:99-122) */
static float atm_density(const TgHipMedium *m, float h, float t0)
{
    const float s = m->falloff_scale, radius = m->falloff_dir[0];
    return expf(-(s*s)*(h*h - radius*radius + t0*t0));
}
static float atm_densityIntegral(const TgHipMedium *m, float h, float t0, float t1)
{
    const float s = m->falloff_scale, radius = m->falloff_dir[0];
    if (t1 == INFINITY)
        return (O_SQRT_PI*0.5f/s)*expf((-h*h + radius*radius)*s*s)*erf_erfc(s*t0);
    else
        return (O_SQRT_PI*0.5f/s)*expf((-h*h + radius*radius)*s*s)*erf_erfDifference(s*t0, s*t1);
}
static float atm_inverseOpticalDepth(const TgHipMedium *m, double h, double t0, double tau)
{
    double s = m->falloff_scale, radius = m->falloff_dir[0];
    double inner = erf(s*t0) + 2.0*(double)O_INV_SQRT_PI*exp(s*s*(h - radius)*(h + radius))*s*tau;
    if (inner >= 1.0)
        return INFINITY;
    else
        return (float)(erf_inv(inner)/s);
}
/* AtmosphericMedium::sampleDistance (AtmosphericMedium.cpp:124-168); exponential transmittance only (include/tungsten_hip.h) */
static int atm_sampleDistance(const TgHipMedium *m, int medium, Sampler *smp, const Ray *ray, MediumState *state, MediumSample *ms)
{
    if (state->bounce > m->max_bounce)
        return 0;
    const v3 sigmaT = ld3(m->sigma_t);
    v3 p = vsub(ray->o, ld3(m->unit_point));
    float t0 = vdot(p, ray->d);
    float h = vlen(vsub(p, vscale(ray->d, t0)));
    float maxT = ray->tmax + t0;
    if (m->absorption_only) {
        ms->t = ray->tmax;
        v3 tau = vscale(sigmaT, atm_densityIntegral(m, h, t0, maxT));
        ms->weight = trans_eval(m, tau, state->firstScatter, 1);
        ms->exited = 1;
    } else {
        int component = (int)(nextSupplemental(smp)*3);           /* sampler.nextDiscrete(3) */
        float sigmaTc = component == 0 ? sigmaT.x : component == 1 ? sigmaT.y : sigmaT.z;
        float tauC = trans_sample(m, smp, state->firstScatter)/sigmaTc;
        float t = atm_inverseOpticalDepth(m, h, t0, tauC);
        ms->t = fminf(t, maxT);
        v3 tau = vscale(sigmaT, atm_densityIntegral(m, h, t0, ms->t));
        ms->exited = t >= maxT;
        ms->weight = trans_eval(m, tau, state->firstScatter, ms->exited);      /* (the exponential transmittance does not look at the flags) */
        float pdf;
        if (ms->exited) {
            pdf = vavg(trans_kernel3(m, state->firstScatter ? 0 : 2, tau));
        } else {
            float rho = atm_density(m, h, ms->t);
            pdf = vavg(vmul(vscale(sigmaT, rho), trans_kernel3(m, state->firstScatter ? 1 : 3, tau)));
            ms->weight = vmul(ms->weight, vscale(vscale(ld3(m->sigma_s), rho), trans_sigmaBar(m)));
        }
        ms->weight = vdivs(ms->weight, pdf);
        ms->t -= t0;
        state->firstScatter = 0; state->bounce++;
    }
    ms->p = vadd(ray->o, vscale(ray->d, ms->t));
    ms->medium = medium;
    return 1;
}

static int medium_sampleDistance(const TgHipSceneDesc *s, int medium, Sampler *smp, const Ray *ray, MediumState *state, MediumSample *ms)
{
    const TgHipMedium *m = &s->media[medium];
    if (m->medium_type == TGHIP_MEDIUM_EXPONENTIAL)
        return expmed_sampleDistance(m, medium, smp, ray, state, ms);
    if (m->medium_type == TGHIP_MEDIUM_ATMOSPHERE)
        return atm_sampleDistance(m, medium, smp, ray, state, ms);
    if (state->bounce > m->max_bounce)
        return 0;
    float maxT = ray->tmax;
    v3 sigmaT = ld3(m->sigma_t);
    if (m->absorption_only) {
        if (maxT == INFINITY)
            return 0;
        ms->t = maxT;
        ms->weight = trans_eval(m, vscale(sigmaT, ms->t), state->firstScatter, 1);
        ms->exited = 1;
    } else {
        int component = (int)(nextSupplemental(smp)*3);           /* sampler.nextDiscrete(3) */
        float sigmaTc = component == 0 ? sigmaT.x : component == 1 ? sigmaT.y : sigmaT.z;
        float t = trans_sample(m, smp, state->firstScatter)/sigmaTc;
        ms->t = fminf(t, maxT);
        ms->exited = t >= maxT;
        v3 tau = vscale(sigmaT, ms->t);
        ms->weight = trans_eval(m, tau, state->firstScatter, ms->exited);
        float pdf;
        if (ms->exited) {
            pdf = vavg(trans_kernel3(m, state->firstScatter ? 0 : 2, tau));               /* surfaceProbability(tau, firstScatter).avg() */
        } else {
            pdf = vavg(vmul(sigmaT, trans_kernel3(m, state->firstScatter ? 1 : 3, tau)));  /* (sigmaT*mediumPdf(tau, firstScatter)).avg() */
            ms->weight = vmul(ms->weight, vscale(ld3(m->sigma_s), trans_sigmaBar(m)));
        }
        ms->weight = vdivs(ms->weight, pdf);
        state->firstScatter = 0; state->bounce++;                  /* state.advance() */
    }
    ms->p = vadd(ray->o, vscale(ray->d, ms->t));
    ms->medium = medium;
    return 1;
}

/* HomogeneousMedium::transmittance (:109-116) */
static v3 medium_transmittance(const TgHipSceneDesc *s, int medium, const Ray *ray, float farT, int startOnSurface, int endOnSurface)
{
    const TgHipMedium *m = &s->media[medium];
    if (m->medium_type == TGHIP_MEDIUM_EXPONENTIAL) {             /* ExponentialMedium::transmittance (ExponentialMedium.cpp:151-163) */
        float x = m->falloff_scale*vdot(vsub(ray->o, ld3(m->unit_point)), ld3(m->falloff_dir));
        float dx = m->falloff_scale*vdot(ray->d, ld3(m->falloff_dir));
        if (farT == INFINITY && dx <= 0.0f)
            return vs(0.0f);
        return trans_eval(m, vscale(ld3(m->sigma_t), expmed_densityIntegral(x, dx, farT)), startOnSurface, endOnSurface);
    }
    if (m->medium_type == TGHIP_MEDIUM_ATMOSPHERE) {              /* AtmosphericMedium::transmittance (AtmosphericMedium.cpp:170-180) */
        v3 p = vsub(ray->o, ld3(m->unit_point));
        float t0 = vdot(p, ray->d);
        float t1 = farT + t0;
        float h = vlen(vsub(p, vscale(ray->d, t0)));
        return trans_eval(m, vscale(ld3(m->sigma_t), atm_densityIntegral(m, h, t0, t1)), startOnSurface, endOnSurface);
    }
    if (farT == INFINITY)
        return vs(0.0f);
    return trans_eval(&s->media[medium], vscale(ld3(s->media[medium].sigma_t), farT), startOnSurface, endOnSurface);
}

/* PhaseFunction::eval / pdf / sample (phasefunctions/IsotropicPhaseFunction.cpp:17-41, HenyeyGreensteinPhaseFunction.cpp:16-63) */
static float phase_hg(float g, float cosTheta)
{
    float term = 1.0f + g*g - 2.0f*g*cosTheta;
    return O_INV_FOUR_PI*(1.0f - g*g)/(term*sqrtf(term));
}
static float phase_rayleigh(float cosTheta) { return (3.0f/(16.0f*O_PI))*(1.0f + cosTheta*cosTheta); }   /* RayleighPhaseFunction.cpp:14-17 */
static float phase_eval(const TgHipMedium *m, v3 wi, v3 wo)   /* eval == pdf for every phase function */
{
    if (m->phase_type == TGHIP_PHASE_RAYLEIGH)
        return phase_rayleigh(vdot(wi, wo));
    if (m->phase_type == TGHIP_PHASE_HENYEY_GREENSTEIN)
        return phase_hg(m->phase_g, vdot(wi, wo));
    return O_INV_FOUR_PI;
}
static void phase_sample(const TgHipMedium *m, Sampler *smp, v3 wi, v3 *w, float *pdf)   /* weight = 1 */
{
    float xi0 = next1D(smp), xi1 = next1D(smp);                    /* next2D */
    float g = m->phase_g;
    if (m->phase_type == TGHIP_PHASE_RAYLEIGH) {                   /* RayleighPhaseFunction::sample (:31-49) */
        float phi = xi0*O_TWO_PI;
        float z = xi1*4.0f - 2.0f;
        float invZ = sqrtf(z*z + 1.0f);
        float u = cbrtf(z + invZ);
        float cosTheta = u - 1.0f/u;
        float sinTheta = sqrtf(fmaxf(1.0f - cosTheta*cosTheta, 0.0f));
        Frame f = frame_from_normal(wi);
        *w = toGlobal(&f, V(cosf(phi)*sinTheta, sinf(phi)*sinTheta, cosTheta));
        *pdf = phase_rayleigh(cosTheta);
    } else if (m->phase_type != TGHIP_PHASE_HENYEY_GREENSTEIN || g == 0.0f) {
        *w = uniformSphere(xi0, xi1);
        *pdf = O_INV_FOUR_PI;
    } else {
        float phi = xi0*O_TWO_PI;
        float cosTheta = (1.0f + g*g - sqr((1.0f - g*g)/(1.0f + g*(xi1*2.0f - 1.0f))))/(2.0f*g);
        float sinTheta = sqrtf(fmaxf(1.0f - cosTheta*cosTheta, 0.0f));
        Frame f = frame_from_normal(wi);
        *w = toGlobal(&f, V(cosf(phi)*sinTheta, sinf(phi)*sinTheta, cosTheta));
        *pdf = phase_hg(g, cosTheta);
    }
}

/* Embree's slab test of ONE child box of a BVH4 node as the SSE4.2 single-ray traversal makes it (kernels/bvh/bvh_intersector_node.h:162-195,
 * TravRay :30-47): rdir = rcp(zero_fix(dir)) with rcp = RCPPS + one Newton step (common/math/vec3fa.h:122-145), planes (bound - org)*rdir,
 * near / far by the sign of rdir, maxi / mini and the final comparison on the floats' bit patterns as signed integers.  The user-geometry
 * BVH has one primitive per leaf (object_accel_max_leaf_size = 1, kernels/common/state.cpp:83-84), so the box is the primitive's own. */
static int embree_maxi(float a, float b) { int32_t ia, ib; memcpy(&ia, &a, 4); memcpy(&ib, &b, 4); return ia > ib; }
static int embree_box_visible(const Ray *ray, v3 lo, v3 hi)
{
    const float o[3] = {ray->o.x, ray->o.y, ray->o.z}, d[3] = {ray->d.x, ray->d.y, ray->d.z};
    const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    float tNear = fmaxf(ray->tmin, 0.0f), tFar = fmaxf(ray->tmax, 0.0f);
    float n[3], f[3];
    for (int k = 0; k < 3; ++k) {
        float a = fabsf(d[k]) < 1e-18f ? 1e-18f : d[k];           /* zero_fix */
        float rdir = embree_rcp(a);
        n[k] = ((rdir >= 0.0f ? l[k] : h[k]) - o[k])*rdir;
        f[k] = ((rdir >= 0.0f ? h[k] : l[k]) - o[k])*rdir;
    }
    /* maxi(maxi(x, y), maxi(z, tnear)), mini likewise */
    float nxy = embree_maxi(n[0], n[1]) ? n[0] : n[1], nzt = embree_maxi(n[2], tNear) ? n[2] : tNear;
    float near = embree_maxi(nxy, nzt) ? nxy : nzt;
    float fxy = embree_maxi(f[0], f[1]) ? f[1] : f[0], fzt = embree_maxi(f[2], tFar) ? tFar : f[2];
    float far = embree_maxi(fxy, fzt) ? fzt : fxy;
    return !embree_maxi(near, far);
}

/* TraceBase::generalizedShadowRay (TraceBase.cpp:62-125).  `endCap` is an object index, `medium` the medium the ray
 * starts in (-1 = none). */
static v3 generalizedShadowRay(Ctx *c, Ray *ray, int medium, int endCap, int startsOnSurface, int bounce)
{
    float initialFarT = ray->tmax;
    v3 throughput = vs(1.0f);
    for (;;) {
        Hit hit;
        Info info;
        c->shadow_rays++;
        int hitAny = scene_intersect(c->s, ray, &hit, c->st);
        int hitObject = -1;
        if (hitAny) {
            intersection_info(c->s, ray, &hit, &info);
            hitObject = info.object;
        }
        if (hitAny && hitObject == endCap && c->s->objects[endCap].type == TGHIP_OBJ_QUAD) {
            /* The light the ray is aimed at is only FOUND when Embree's slab test lets the ray into its box -- and a quad's box is flat: from
             * the second segment of a shadow ray on (new origin, remaining farT) the quad may lie an ulp inside farT by Quad::intersect's
             * arithmetic and an ulp behind it by the slab test's.  Embree then never calls Quad::intersect, info.primitive stays null, the
             * function returns as it would at the end cap -- but ray.farT() was not shortened to the hit, and the medium's transmittance of
             * the segment is taken over the remaining distance.  (Quad::bounds, Quad.cpp:281-289; found with the stress render of the
             * fog + smoke twin, DESIGN.md section 8.  Other light shapes: their boxes are not restated, the light is always found.) */
            const TgHipObject *o = &c->s->objects[endCap];
            v3 b = ld3(o->base), e0 = ld3(o->edge0), e1 = ld3(o->edge1);
            v3 p[4] = {b, vadd(b, e0), vadd(b, e1), vadd(vadd(b, e0), e1)};
            v3 lo = p[0], hi = p[0];
            for (int k = 1; k < 4; ++k) {
                lo = V(fminf(lo.x, p[k].x), fminf(lo.y, p[k].y), fminf(lo.z, p[k].z));
                hi = V(fmaxf(hi.x, p[k].x), fmaxf(hi.y, p[k].y), fmaxf(hi.z, p[k].z));
            }
            if (!embree_box_visible(ray, lo, hi))
                hitAny = 0;
        }
        int didHit = hitAny && hitObject != endCap;
        if (didHit) {
            if (!(c->s->bsdfs[info.bsdf].lobes & TGHIP_LOBE_FORWARD))
                return vs(0.0f);
            Local l = makeLocalScatterEvent(c, &info, ray);
            Event fe = make_event(c, &info, &l, TGHIP_LOBE_FORWARD);   /* makeForwardEvent */
            fe.wo = vneg(fe.wi);
            v3 transparency = bsdf_eval_rt(c->s, info.bsdf, &fe);
            if (viszero(transparency))
                return vs(0.0f);
            throughput = vmul(throughput, transparency);
            bounce++;
            if (bounce >= c->s->settings.max_bounces)
                return vs(0.0f);
        }
        if (medium >= 0)                                   /* :103-112; ray.farT() is the hit distance when anything was hit */
            throughput = vmul(throughput, medium_transmittance(c->s, medium, ray, hitAny ? hit.t : ray->tmax, startsOnSurface, 1));   /* endsOnSurface = true (attenuatedEmission) */
        if (!hitAny || hitObject == endCap)
            return bounce >= c->s->settings.min_bounces ? throughput : vs(0.0f);
        medium = selectMedium(&c->s->objects[info.object], medium, !info.backSide);     /* :115 */
        startsOnSurface = 1;
        ray->o = vadd(ray->o, vscale(ray->d, hit.t));      /* ray.hitpoint(): farT was set to the hit */
        initialFarT -= hit.t;
        ray->tmin = info.epsilon;
        ray->tmax = initialFarT;
    }
}

/* light.intersect + intersectionInfo + evalDirect for the light kinds in scope.
 * Returns 0 if the ray misses the light.  (Quad.cpp:71-131,235-238; InfiniteSphere.cpp:77-104,241-244) */
typedef struct { float t, u, v; int backSide; v3 w; v3 n; v3 o; } LightHit;   /* n: surface normal (cube), o: ray origin */
static int light_intersect(const TgHipSceneDesc *s, int objIdx, const Ray *ray, LightHit *lh)
{
    const TgHipObject *o = &s->objects[objIdx];
    lh->w = ray->d;
    lh->o = ray->o;
    lh->n = V(0.0f, 0.0f, 0.0f);
    if (o->type == TGHIP_OBJ_QUAD) {
        TgHipPrimRec r;
        memset(&r, 0, sizeof(r));
        memcpy(r.a, o->base, 12); memcpy(r.b, o->edge0, 12); memcpy(r.c, o->edge1, 12);
        r.p0 = o->inv_uv_sq[0]; r.p1 = o->inv_uv_sq[1];
        return quad_test(&r, o, ray, ray->tmax, &lh->t, &lh->u, &lh->v, &lh->backSide);
    } else if (o->type == TGHIP_OBJ_INFINITE_SPHERE) {
        lh->t = ray->tmax; lh->backSide = 0;
        inf_directionToUV(o, ray->d, &lh->u, &lh->v, NULL);
        return 1;
    } else if (o->type == TGHIP_OBJ_INFINITE_SPHERE_CAP) {     /* InfiniteSphereCap::intersect + intersectionInfo (:60-90) */
        if (vdot(ray->d, ld3(o->normal)) < o->scale[0]) return 0;
        lh->t = ray->tmax; lh->backSide = 0; lh->u = lh->v = 0.0f;
        return 1;
    } else if (o->type == TGHIP_OBJ_CUBE) {            /* Cube::intersect + intersectionInfo */
        if (!cube_test(o, ray, ray->tmax, &lh->t, &lh->backSide)) return 0;
        cube_surface(o, vadd(ray->o, vscale(ray->d, lh->t)), &lh->n, &lh->u, &lh->v);
        return 1;
    } else if (o->type == TGHIP_OBJ_SPHERE) {          /* Sphere::intersect + intersectionInfo */
        if (!sphere_test(o, ray, ray->tmax, &lh->t, &lh->backSide)) return 0;
        sphere_surface(o, vadd(ray->o, vscale(ray->d, lh->t)), &lh->n, &lh->u, &lh->v);
        return 1;
    } else if (o->type == TGHIP_OBJ_CYLINDER) {        /* Cylinder::intersect + intersectionInfo */
        float cap;
        if (!cylinder_test(o, ray, ray->tmax, &lh->t, &lh->backSide, &cap)) return 0;
        cylinder_surface(o, ray, lh->t, cap, &lh->n, &lh->u, &lh->v);
        return 1;
    } else if (o->type == TGHIP_OBJ_DISK) {            /* Disk::intersect + intersectionInfo */
        float rSq;
        if (!disk_test(o, ray, ray->tmax, &lh->t, &rSq, &lh->backSide)) return 0;
        disk_surface(o, vadd(ray->o, vscale(ray->d, lh->t)), rSq, &lh->u, &lh->v);
        lh->n = ld3(o->normal);
        return 1;
    } else if (o->type == TGHIP_OBJ_MESH) {            /* TriangleMesh::intersect + intersectionInfo: the mesh's own BVH */
        Hit hit;
        if (!scene_intersect_obj(s, ray, &hit, NULL, objIdx)) return 0;
        Info info;
        intersection_info(s, ray, &hit, &info);
        lh->t = hit.t; lh->u = info.u; lh->v = info.v; lh->backSide = info.backSide; lh->n = info.Ng;
        return 1;
    }
    return 0;
}
static v3 light_evalDirect(const TgHipSceneDesc *s, int objIdx, const LightHit *lh)
{
    const TgHipObject *o = &s->objects[objIdx];
    if (o->emission < 0) return vs(0.0f);
    if (lh->backSide) return vs(0.0f);
    return texture_eval(s, o->emission, lh->u, lh->v);
}
/* Quad::directPdf (Quad.cpp:216-223), InfiniteSphere::directPdf (InfiniteSphere.cpp:218-229) */
static float light_directPdf(const TgHipSceneDesc *s, int objIdx, const LightHit *lh, v3 p)
{
    const TgHipObject *o = &s->objects[objIdx];
    if (o->type == TGHIP_OBJ_QUAD) {
        v3 n = ld3(o->normal);
        float cosTheta = fabsf(vdot(n, lh->w));
        float t = vdot(n, vsub(ld3(o->base), p))/vdot(n, lh->w);
        return t*t/(cosTheta*o->area);
    } else if (o->type == TGHIP_OBJ_CUBE) {            /* Cube.cpp:291-295 */
        v3 hp = vadd(lh->o, vscale(lh->w, lh->t));
        return vlensq(vsub(p, hp))/(-vdot(lh->w, lh->n)*o->area);
    } else if (o->type == TGHIP_OBJ_MESH) {            /* TriangleMesh.cpp:469-473 */
        v3 hp = vadd(lh->o, vscale(lh->w, lh->t));
        return vlensq(vsub(p, hp))/(-vdot(lh->w, lh->n)*o->area);
    } else if (o->type == TGHIP_OBJ_CYLINDER) {        /* Cylinder::directPdf (Cylinder.cpp:246-250) */
        v3 hp = vadd(lh->o, vscale(lh->w, lh->t));
        return vlensq(vsub(p, hp))/(-vdot(lh->w, lh->n)*o->area);
    } else if (o->type == TGHIP_OBJ_POINT) {           /* Point::directPdf (Point.cpp:117-121) */
        return vlensq(vsub(p, ld3(o->pos)));
    } else if (o->type == TGHIP_OBJ_INFINITE_SPHERE_CAP) {     /* :214-218: uniformSphericalCapPdf */
        return O_INV_TWO_PI/(1.0f - o->scale[0]);
    } else if (o->type == TGHIP_OBJ_DISK) {            /* Disk::directPdf (Disk.cpp:228-235) */
        v3 n = ld3(o->normal);
        float cosTheta = fabsf(vdot(n, lh->w));
        float t = vdot(n, vsub(ld3(o->pos), p))/vdot(n, lh->w);
        return t*t/(cosTheta*o->scale[0]*o->scale[0]*O_PI);
    } else if (o->type == TGHIP_OBJ_SPHERE) {          /* Sphere.cpp:216-222 */
        float dist = vlen(vsub(ld3(o->pos), p));
        float cosTheta = sqrtf(fmaxf(dist*dist - o->scale[0]*o->scale[0], 0.0f))/dist;
        return O_INV_TWO_PI/(1.0f - cosTheta);         /* SampleWarp::uniformSphericalCapPdf */
    } else {
        const TgHipTexture *t = &s->textures[o->emission];
        if (t->type != TGHIP_TEX_BITMAP)       /* _emission->isConstant() (checker envmaps are outside the scope) */
            return O_INV_FOUR_PI;
        float sinTheta, u, v;
        inf_directionToUV(o, lh->w, &u, &v, &sinTheta);
        return O_INV_PI*O_INV_TWO_PI*bitmap_pdf(s, t, u, v)/sinTheta;
    }
}
/* Quad::sampleDirect (Quad.cpp:172-187), InfiniteSphere::sampleDirect (InfiniteSphere.cpp:161-176) */
static int light_sampleDirect(const TgHipSceneDesc *s, int objIdx, v3 p, Sampler *smp, v3 *d, float *dist, float *pdf)
{
    const TgHipObject *o = &s->objects[objIdx];
    if (o->type == TGHIP_OBJ_QUAD) {
        v3 n = ld3(o->normal);
        if (vdot(n, vsub(p, ld3(o->base))) <= 0.0f)
            return 0;
        float xi0 = next1D(smp), xi1 = next1D(smp);
        v3 q = vadd(vadd(ld3(o->base), vscale(ld3(o->edge0), xi0)), vscale(ld3(o->edge1), xi1));
        v3 dd = vsub(q, p);
        float rSq = vlensq(dd);
        *dist = sqrtf(rSq);
        dd = vdivs(dd, *dist);
        float cosTheta = -vdot(n, dd);
        *pdf = rSq/(cosTheta*o->area);
        *d = dd;
        return 1;
    } else if (o->type == TGHIP_OBJ_CUBE) {            /* Cube::sampleDirect / samplePosition / sampleFace (Cube.cpp:229-245, 189-213, 42-55) */
        float u = next1D(smp);
        float uOrig = u;
        int dim;
        u *= o->face_cdf[2];
        if (u < o->face_cdf[0]) { u /= o->face_cdf[0]; dim = 0; }
        else if (u < o->face_cdf[1]) { u = (u - o->face_cdf[0])/(o->face_cdf[1] - o->face_cdf[0]); dim = 1; }
        else { u = (u - o->face_cdf[1])/(o->face_cdf[2] - o->face_cdf[1]); dim = 2; }
        (void)uOrig;
        int sAx = (dim + 1) % 3, tAx = (dim + 2) % 3;
        float xi0 = next1D(smp), xi1 = next1D(smp);
        float nn[3] = {0.0f, 0.0f, 0.0f}, pp[3] = {0.0f, 0.0f, 0.0f};
        nn[dim] = u < 0.5f ? -1.0f : 1.0f;             /* u is the value sampleFace left behind (it takes u by reference) */
        pp[dim] = nn[dim]*o->scale[dim];
        pp[sAx] = (xi0*2.0f - 1.0f)*o->scale[sAx];
        pp[tAx] = (xi1*2.0f - 1.0f)*o->scale[tAx];
        v3 q = vadd(mat3_mul(o->rot, V(pp[0], pp[1], pp[2])), ld3(o->pos));
        v3 Ng = mat3_mul(o->rot, V(nn[0], nn[1], nn[2]));
        v3 L = vsub(q, p);
        float rSq = vlensq(L);
        *dist = sqrtf(rSq);
        *d = vdivs(L, *dist);
        float cosTheta = -vdot(Ng, *d);
        if (cosTheta <= 0.0f)
            return 0;
        *pdf = rSq/(cosTheta*o->area);
        return 1;
    } else if (o->type == TGHIP_OBJ_MESH) {            /* TriangleMesh::sampleDirect / samplePosition (TriangleMesh.cpp:411-462) */
        const float *cdf = s->light_tris + o->first_light_tri;
        const float *tris = cdf + o->num_light_tris + 1;
        float u = next1D(smp);
        int idx = upper_bound_idx(cdf, o->num_light_tris + 1, u) - 1;     /* Distribution1D::warp */
        const float *t = tris + (size_t)idx*9;
        v3 p0 = ld3(t), p1 = ld3(t + 3), p2 = ld3(t + 6);
        v3 normal = vnorm(vcross(vsub(p1, p0), vsub(p2, p0)));
        float xi0 = next1D(smp), xi1 = next1D(smp);
        float uSqrt = sqrtf(xi0);                                           /* SampleWarp::uniformTriangleUv */
        float alpha = 1.0f - uSqrt, beta = (1.0f - xi1)*uSqrt;
        v3 q = vadd(vadd(vscale(p0, alpha), vscale(p1, beta)), vscale(p2, 1.0f - alpha - beta));
        v3 L = vsub(q, p);
        float rSq = vlensq(L);
        *dist = sqrtf(rSq);
        *d = vdivs(L, *dist);
        float cosTheta = -vdot(normal, *d);
        if (cosTheta <= 0.0f)
            return 0;
        *pdf = rSq/(cosTheta*o->area);
        return 1;
    } else if (o->type == TGHIP_OBJ_POINT) {           /* Point::sampleDirect (Point.cpp:93-101): no random numbers */
        v3 L = vsub(ld3(o->pos), p);
        float rSq = vlensq(L);
        *dist = sqrtf(rSq);
        *d = vdivs(L, *dist);
        *pdf = rSq;
        return 1;
    } else if (o->type == TGHIP_OBJ_INFINITE_SPHERE_CAP) {     /* InfiniteSphereCap::sampleDirect (:130-138) */
        float xi0 = next1D(smp), xi1 = next1D(smp);
        float phi = xi0*O_TWO_PI;                      /* SampleWarp::uniformSphericalCap */
        float z = xi1*(1.0f - o->scale[0]) + o->scale[0];
        float r = sqrtf(fmaxf(1.0f - z*z, 0.0f));
        v3 local = V(cosf(phi)*r, sinf(phi)*r, z);
        Frame frame = {ld3(o->normal), ld3(o->edge0), ld3(o->edge1)};
        *d = toGlobal(&frame, local);
        *dist = INFINITY;
        *pdf = O_INV_TWO_PI/(1.0f - o->scale[0]);
        return 1;
    } else if (o->type == TGHIP_OBJ_CYLINDER) {        /* Cylinder::sampleDirect + samplePosition (Cylinder.cpp:149-170, 181-196) */
        const float radius = o->scale[0], halfHeight = o->scale[1];
        v3 ng, q;
        if (o->scale[2] != 0.0f && nextBoolean(smp, O_TWO_PI*sqr(radius)*o->inv_area)) {
            float xi0 = next1D(smp), xi1 = next1D(smp);
            float phi = xi0*O_TWO_PI, rr = sqrtf(xi1);     /* SampleWarp::uniformDisk */
            float sign = nextBoolean(smp, 0.5f) ? -1.0f : 1.0f;
            ng = V(0.0f, sign, 0.0f);
            q = V(cosf(phi)*rr*radius, sign*halfHeight, sinf(phi)*rr*radius);
        } else {
            float xi0 = next1D(smp), xi1 = next1D(smp);
            float phi = xi0*O_TWO_PI;                      /* SampleWarp::uniformCylinder */
            float cx = cosf(phi), cy = sinf(phi), cz = xi1*2.0f - 1.0f;
            ng = V(cx, 0.0f, cy);
            q = V(cx*radius, cz*halfHeight, cy*radius);
        }
        ng = mat3_mul(o->rot, ng);
        q = vadd(mat3_mul(o->rot, q), ld3(o->pos));
        v3 L = vsub(q, p);
        float rSq = vlensq(L);
        *dist = sqrtf(rSq);
        *d = vdivs(L, *dist);
        float cosTheta = -vdot(ng, *d);
        if (cosTheta <= 0.0f)
            return 0;
        *pdf = rSq/(cosTheta*o->area);
        return 1;
    } else if (o->type == TGHIP_OBJ_DISK) {            /* Disk::sampleDirect (Disk.cpp:178-194) */
        v3 n = ld3(o->normal), center = ld3(o->pos);
        if (vdot(n, vsub(p, center)) < 0.0f)
            return 0;
        float xi0 = next1D(smp), xi1 = next1D(smp);
        float phi = xi0*O_TWO_PI, rr = sqrtf(xi1);     /* SampleWarp::uniformDisk */
        float lx = cosf(phi)*rr*o->scale[0], ly = sinf(phi)*rr*o->scale[0];
        v3 q = vadd(vadd(center, vscale(ld3(o->edge1), lx)), vscale(ld3(o->edge0), ly));   /* lQ.x*bitangent + lQ.y*tangent */
        v3 L = vsub(q, p);
        float rSq = vlensq(L);
        *dist = sqrtf(rSq);
        *d = vdivs(L, *dist);
        if (-vdot(*d, n) < o->scale[1])
            return 0;
        float cosTheta = -vdot(n, *d);
        *pdf = rSq/(cosTheta*o->scale[0]*o->scale[0]*O_PI);
        return 1;
    } else if (o->type == TGHIP_OBJ_SPHERE) {          /* Sphere::sampleDirect (Sphere.cpp:173-194) */
        v3 L = vsub(ld3(o->pos), p);
        float dd = vlen(L);
        float C = dd*dd - o->scale[0]*o->scale[0];
        if (C <= 0.0f)
            return 0;
        L = vnorm(L);
        float cosTheta = sqrtf(C)/dd;
        float xi0 = next1D(smp), xi1 = next1D(smp);
        float phi = xi0*O_TWO_PI;                      /* SampleWarp::uniformSphericalCap (SampleWarp.hpp:119-129) */
        float z = xi1*(1.0f - cosTheta) + cosTheta;
        float r = sqrtf(fmaxf(1.0f - z*z, 0.0f));
        v3 local = V(cosf(phi)*r, sinf(phi)*r, z);
        float B = dd*local.z;
        float det = sqrtf(fmaxf(B*B - C, 0.0f));
        *dist = B - det;
        Frame frame = frame_from_normal(L);
        *d = toGlobal(&frame, local);
        *pdf = O_INV_TWO_PI/(1.0f - cosTheta);
        return 1;
    } else {
        const TgHipTexture *t = &s->textures[o->emission];
        float xi0 = next1D(smp), xi1 = next1D(smp);
        if (t->type != TGHIP_TEX_BITMAP) {
            *d = uniformSphere(xi0, xi1);
            *dist = INFINITY;
            *pdf = O_INV_FOUR_PI;
            return 1;
        }
        float u, v, sinTheta;
        bitmap_sample(s, t, xi0, xi1, &u, &v);
        *d = inf_uvToDirection(o, u, v, &sinTheta);
        *pdf = O_INV_PI*O_INV_TWO_PI*bitmap_pdf(s, t, u, v)/sinTheta;
        *dist = INFINITY;
        return *pdf != 0.0f;
    }
}
/* Quad::approximateRadiance (Quad.cpp:256-279), InfiniteSphere::approximateRadiance (InfiniteSphere.cpp:261-266) */
static float light_approximateRadiance(const TgHipSceneDesc *s, int objIdx, v3 p)
{
    const TgHipObject *o = &s->objects[objIdx];
    if (o->type == TGHIP_OBJ_QUAD) {
        if (o->emission < 0) return 0.0f;
        v3 R0 = vsub(ld3(o->base), p);
        if (vdot(R0, ld3(o->normal)) >= 0.0f)
            return 0.0f;
        v3 R1 = vadd(R0, ld3(o->edge0));
        v3 R2 = vadd(R1, ld3(o->edge1));
        v3 R3 = vadd(R0, ld3(o->edge1));
        v3 n0 = vnorm(vcross(R0, R1)), n1 = vnorm(vcross(R1, R2)), n2 = vnorm(vcross(R2, R3)), n3 = vnorm(vcross(R3, R0));
        float Q = acosf(vdot(n0, n1)) + acosf(vdot(n1, n2)) + acosf(vdot(n2, n3)) + acosf(vdot(n3, n0));
        return (O_TWO_PI - fabsf(Q))*vmax3(ld3(s->textures[o->emission].avg));
    } else if (o->type == TGHIP_OBJ_CUBE) {            /* Cube.cpp:326-330 */
        v3 lp = mat3_tmul(o->rot, vsub(p, ld3(o->pos)));
        v3 ap = V(fmaxf(fabsf(lp.x), 0.0f), fmaxf(fabsf(lp.y), 0.0f), fmaxf(fabsf(lp.z), 0.0f));
        float dSq = vlensq(ap);
        return vmax3(ld3(s->textures[o->emission].avg))*o->face_cdf[2]/dSq;
    } else if (o->type == TGHIP_OBJ_MESH || o->type == TGHIP_OBJ_CYLINDER) {   /* TriangleMesh.cpp:514-517, Cylinder.cpp:280-284: "unknown" */
        return -1.0f;
    } else if (o->type == TGHIP_OBJ_POINT) {           /* Point::approximateRadiance (Point.cpp:166-169) */
        /* scale = Point::_power as prepareForRender left it (Point.cpp:186): 0 for a light given by "power" */
        return O_INV_FOUR_PI*vmax3(ld3(o->scale))/vlensq(vsub(ld3(o->pos), p));
    } else if (o->type == TGHIP_OBJ_INFINITE_SPHERE_CAP) {     /* :220-225 */
        if (o->emission < 0 || !(o->flags & TGHIP_OBJF_SAMPLE)) return 0.0f;
        return O_TWO_PI*(1.0f - o->scale[0])*vmax3(ld3(s->textures[o->emission].avg));
    } else if (o->type == TGHIP_OBJ_DISK) {            /* Disk::approximateRadiance (Disk.cpp:253-281) */
        if (o->emission < 0) return 0.0f;
        v3 n = ld3(o->normal);
        v3 coneD = vsub(p, ld3(o->base));
        if (vdot(coneD, n)/vlen(coneD) < o->scale[1])
            return 0.0f;
        v3 dd = vsub(ld3(o->pos), p);
        v3 e0 = vscale(ld3(o->edge0), o->scale[0]), e1 = vscale(ld3(o->edge1), o->scale[0]);
        v3 R0 = vsub(vsub(dd, e0), e1);
        v3 R1 = vadd(R0, vscale(e0, 2.0f));
        v3 R2 = vadd(R1, vscale(e1, 2.0f));
        v3 R3 = vadd(R0, vscale(e1, 2.0f));
        v3 n0 = vnorm(vcross(R0, R1)), n1 = vnorm(vcross(R1, R2)), n2 = vnorm(vcross(R2, R3)), n3 = vnorm(vcross(R3, R0));
        float Q = acosf(vdot(n0, n1)) + acosf(vdot(n1, n2)) + acosf(vdot(n2, n3)) + acosf(vdot(n3, n0));
        return (O_TWO_PI - fabsf(Q))*vmax3(ld3(s->textures[o->emission].avg));
    } else if (o->type == TGHIP_OBJ_SPHERE) {          /* Sphere.cpp:266-271, 33-40 */
        if (o->emission < 0) return 0.0f;
        v3 L = vsub(ld3(o->pos), p);
        float dd = vlen(L);
        float cosTheta = sqrtf(fmaxf(dd*dd - o->scale[0]*o->scale[0], 0.0f))/dd;
        return O_TWO_PI*(1.0f - cosTheta)*vmax3(ld3(s->textures[o->emission].avg));
    } else {
        if (o->emission < 0 || !(o->flags & TGHIP_OBJF_SAMPLE)) return 0.0f;
        if (o->flags & TGHIP_OBJF_SKYDOME)              /* Skydome::approximateRadiance (Skydome.cpp:240-243) */
            return O_FOUR_PI*vmax3(ld3(s->textures[o->emission].avg));
        return O_TWO_PI*vmax3(ld3(s->textures[o->emission].avg));
    }
}

/* TraceBase::attenuatedEmission (TraceBase.cpp:144-174) for non-Dirac lights */
static v3 attenuatedEmission(Ctx *c, int lightObj, int medium, float expectedDist, int bounce, int startsOnSurface, Ray *ray, LightHit *lh)
{
    const float fudgeFactor = 1.0f + 1e-3f;
    if (c->s->objects[lightObj].type == TGHIP_OBJ_POINT) {       /* light.isDirac(): ray.setFarT(expectedDist) (:157-158) */
        lh->t = expectedDist; lh->u = lh->v = 0.0f; lh->backSide = 0;
    } else if (!light_intersect(c->s, lightObj, ray, lh) || lh->t*fudgeFactor < expectedDist) {
        return vs(0.0f);
    }
    ray->tmax = lh->t;
    v3 shadow = generalizedShadowRay(c, ray, medium, lightObj, startsOnSurface, bounce);
    if (c->transmittanceOut)                               /* if (transmittance) *transmittance = shadow (:169-170) */
        *c->transmittanceOut = shadow;
    if (viszero(shadow))
        return vs(0.0f);
    return vmul(shadow, light_evalDirect(c->s, lightObj, lh));
}

/* TraceBase::lightSample (TraceBase.cpp:246-285) */
static v3 lightSample(Ctx *c, int lightObj, const Info *info, const Local *l, int medium, int bounce)
{
    v3 d; float dist, pdf;
    if (!light_sampleDirect(c->s, lightObj, info->p, c->sampler, &d, &dist, &pdf))
        return vs(0.0f);
    Event e = make_event(c, info, l, LOBE_ALL_BUT_SPECULAR);
    e.wo = toLocal(&l->frame, d);
    if (!isConsistent(c, info, l, e.wo, d))
        return vs(0.0f);
    v3 f = bsdf_eval_rt(c->s, info->bsdf, &e);
    if (viszero(f))
        return vs(0.0f);
    medium = selectMedium(&c->s->objects[info->object], medium, vdot(d, info->Ng) < 0.0f);   /* :260-261 */
    Ray ray = {info->p, d, info->epsilon, INFINITY};
    LightHit lh;
    c->transmittanceOut = c->visRequest;
    v3 em = attenuatedEmission(c, lightObj, medium, dist, bounce, 1, &ray, &lh);
    c->transmittanceOut = NULL;
    if (viszero(em))
        return vs(0.0f);
    v3 lightF = vdivs(vmul(f, em), pdf);
    if (c->s->objects[lightObj].type != TGHIP_OBJ_POINT)         /* !light.isDirac() (:281-282) */
        lightF = vscale(lightF, powerHeuristic(pdf, bsdf_pdf(c->s, info->bsdf, &e)));
    return lightF;
}

/* TraceBase::bsdfSample (TraceBase.cpp:287-321) */
static v3 bsdfSample(Ctx *c, int lightObj, const Info *info, const Local *l, int medium, int bounce)
{
    Event e = make_event(c, info, l, LOBE_ALL_BUT_SPECULAR);
    if (!bsdf_sample_rt(c->s, info->bsdf, &e))
        return vs(0.0f);
    if (viszero(e.weight))
        return vs(0.0f);
    v3 wo = toGlobal(&l->frame, e.wo);
    if (!isConsistent(c, info, l, e.wo, wo))
        return vs(0.0f);
    medium = selectMedium(&c->s->objects[info->object], medium, vdot(wo, info->Ng) < 0.0f);  /* :302-303 */
    Ray ray = {info->p, wo, info->epsilon, INFINITY};
    LightHit lh;
    v3 em = attenuatedEmission(c, lightObj, medium, -1.0f, bounce, 1, &ray, &lh);
    if (viszero(em))
        return vs(0.0f);
    v3 bsdfF = vmul(em, e.weight);
    bsdfF = vscale(bsdfF, powerHeuristic(e.pdf, light_directPdf(c->s, lightObj, &lh, info->p)));
    return bsdfF;
}

/* TraceBase::chooseLight (TraceBase.cpp:416-459); returns the light's object index or -1 */
static int chooseLight(Ctx *c, v3 p, float *weight)
{
    const TgHipSceneDesc *s = c->s;
    int n = (int)s->num_lights;
    if (n == 0) return -1;
    if (n == 1) { *weight = 1.0f; return s->lights[0]; }
    float lightPdf[64];
    if (n > 64) n = 64;
    float total = 0.0f;
    unsigned numNonNegative = 0;
    for (int i = 0; i < n; ++i) {
        lightPdf[i] = light_approximateRadiance(s, s->lights[i], p);
        if (lightPdf[i] >= 0.0f) { total += lightPdf[i]; numNonNegative++; }
    }
    if (numNonNegative == 0) {
        for (int i = 0; i < n; ++i) lightPdf[i] = 1.0f;
        total = (float)n;
    } else if ((int)numNonNegative < n) {
        for (int i = 0; i < n; ++i) {
            float uniformWeight = (total == 0.0f ? 1.0f : total)/numNonNegative;
            if (lightPdf[i] < 0.0f) { lightPdf[i] = uniformWeight; total += uniformWeight; }
        }
    }
    if (total == 0.0f) return -1;
    float t = next1D(c->sampler)*total;
    for (int i = 0; i < n; ++i) {
        if (t < lightPdf[i] || i == n - 1) { *weight = total/lightPdf[i]; return s->lights[i]; }
        t -= lightPdf[i];
    }
    return -1;
}

/* TraceBase::estimateDirect + sampleDirect (TraceBase.cpp:483-494, 383-400) */
static v3 estimateDirect(Ctx *c, const Info *info, const Local *l, int medium, int bounce)
{
    float weight;
    int light = chooseLight(c, info->p, &weight);
    if (light < 0)
        return vs(0.0f);
    uint32_t lobes = c->s->bsdfs[info->bsdf].lobes;
    int pureSpecular = lobes != 0 && (lobes & ~(uint32_t)LOBE_SPECULAR) == 0;
    if (pureSpecular || lobes == TGHIP_LOBE_FORWARD)
        return vs(0.0f);
    v3 result = lightSample(c, light, info, l, medium, bounce);
    if (c->s->objects[light].type != TGHIP_OBJ_POINT)             /* !light.isDirac() (TraceBase.cpp:396-397) */
        result = vadd(result, bsdfSample(c, light, info, l, medium, bounce));
    return vscale(result, weight);
}

/* TraceBase::volumeEstimateDirect + volumeSampleDirect + volumeLightSample + volumePhaseSample (TraceBase.cpp:323-381, 402-414, 471-481) */
static v3 volumeEstimateDirect(Ctx *c, const MediumSample *ms, int medium, int bounce, v3 parentDir)
{
    const TgHipMedium *m = &c->s->media[ms->medium];
    float weight;
    int light = chooseLight(c, ms->p, &weight);
    if (light < 0)
        return vs(0.0f);
    const int dirac = c->s->objects[light].type == TGHIP_OBJ_POINT;
    v3 result = vs(0.0f);
    {   /* volumeLightSample */
        v3 d; float dist, pdf;
        if (light_sampleDirect(c->s, light, ms->p, c->sampler, &d, &dist, &pdf)) {
            float f = phase_eval(m, parentDir, d);
            if (f != 0.0f) {
                Ray ray = {ms->p, d, 0.0f, INFINITY};              /* parentRay.scatter(p, d, 0.0f) */
                LightHit lh;
                v3 e = attenuatedEmission(c, light, medium, dist, bounce, 0, &ray, &lh);
                if (!viszero(e)) {
                    v3 lightF = vdivs(vscale(e, f), pdf);
                    if (!dirac)
                        lightF = vscale(lightF, powerHeuristic(pdf, phase_eval(m, parentDir, d)));
                    result = vadd(result, lightF);
                }
            }
        }
    }
    if (!dirac) {   /* volumePhaseSample */
        v3 w; float pdf;
        phase_sample(m, c->sampler, parentDir, &w, &pdf);
        Ray ray = {ms->p, w, 0.0f, INFINITY};
        LightHit lh;
        v3 e = attenuatedEmission(c, light, medium, -1.0f, bounce, 0, &ray, &lh);
        if (!viszero(e))
            result = vadd(result, vscale(e, powerHeuristic(pdf, light_directPdf(c->s, light, &lh, ms->p))));
    }
    return vscale(result, weight);
}

/* PathTracer::traceSample (PathTracer.cpp:14-149); low_order_scattering and include_surfaces at their defaults */

static v3 traceSample(Ctx *c, uint32_t px, uint32_t py)
{
    const TgHipSceneDesc *s = c->s;
    const TgHipCamera *cam = &s->camera;
    const int maxBounces = s->settings.max_bounces, minBounces = s->settings.min_bounces;
    const int nee = s->settings.enable_light_sampling;

    /* ThinlensCamera::samplePosition (ThinlensCamera.cpp:85-97) draws the lens point first; the aperture is the default
     * DiskTexture: sample = uniformDisk(xi).xy*0.5 + 0.5 (DiskTexture.cpp:78-81, SampleWarp.hpp:64-69) */
    const int thinlens = cam->type == TGHIP_CAMERA_THINLENS;
    v3 lensP = ld3(cam->pos);
    if (thinlens) {
        float l0 = next1D(c->sampler), l1 = next1D(c->sampler);
        float su, sv;
        if (cam->aperture_type == TGHIP_APERTURE_BLADE) {
            /* BladeTexture::sample (textures/BladeTexture.cpp:110-130): a blade, then a point of its triangle */
            float u = l0*(float)cam->blade_count;
            int blade = (int)u;
            u -= (float)blade;
            float phi = cam->blade_angle + (float)blade*cam->blade_step;
            float sinPhi = sinf(phi), cosPhi = cosf(phi);
            float uSqrt = sqrtf(u);
            float alpha = 1.0f - uSqrt, beta = (1.0f - l1)*uSqrt;
            float lx = (1.0f + cam->blade_edge[0])*beta + (1.0f - alpha - beta), ly = cam->blade_edge[1]*beta;
            su = (lx*cosPhi - ly*sinPhi)*0.5f + 0.5f;
            sv = (ly*cosPhi + lx*sinPhi)*0.5f + 0.5f;
        } else if (cam->aperture_type == TGHIP_APERTURE_BITMAP) {
            /* BitmapTexture::sample(MAP_UNIFORM, lensUv) (textures/BitmapTexture.cpp:433-439) on the aperture's own distribution
             * (ThinlensCamera::precompute, cameras/ThinlensCamera.cpp:27-35) */
            Dist2D ad;
            ad.w = cam->aperture_w; ad.h = cam->aperture_h;
            ad.mpdf = s->dist + cam->aperture_dist;
            ad.mcdf = ad.mpdf + ad.h;
            ad.pdf = ad.mcdf + ad.h + 1;
            ad.cdf = ad.pdf + (size_t)ad.w*ad.h;
            int row, column;
            float nu = l0, nv = l1;
            dist_warp(&ad, &nu, &nv, &row, &column);
            su = (nu + column)/ad.w;
            sv = 1.0f - (nv + row)/ad.h;
        } else {
            float phi = l0*O_TWO_PI, r = sqrtf(l1);
            su = cosf(phi)*r*0.5f + 0.5f; sv = sinf(phi)*r*0.5f + 0.5f;
        }
        float ax = su*2.0f - 1.0f, ay = sv*2.0f - 1.0f;
        ax *= cam->aperture_size; ay *= cam->aperture_size;
        /* _transform*Vec3f(ax, ay, 0) (Mat4f::operator*(Vec3f)) */
        lensP = V(cam->xf[0]*ax + cam->xf[1]*ay + cam->xf[2]*0.0f + cam->pos[0],
                  cam->xf[3]*ax + cam->xf[4]*ay + cam->xf[5]*0.0f + cam->pos[1],
                  cam->xf[6]*ax + cam->xf[7]*ay + cam->xf[8]*0.0f + cam->pos[2]);
    }
    /* PinholeCamera::samplePosition/sampleDirection (PinholeCamera.cpp:53-86) */
    float xi0 = next1D(c->sampler), xi1 = next1D(c->sampler);
    float fu, fv;
    if (cam->filter_type == TGHIP_FILTER_DIRAC) { fu = fv = 0.0f; }
    else if (cam->filter_type == TGHIP_FILTER_BOX) { fu = xi0 - 0.5f; fv = xi1 - 0.5f; }
    else {
        /* ReconstructionFilter::sample (ReconstructionFilter.hpp:86-103) */
        float xi[2] = {xi0, xi1}, out[2];
        for (int k = 0; k < 2; ++k) {
            float x = xi[k];
            int negative = x < 0.5f;
            x = negative ? x*2.0f : (x - 0.5f)*2.0f;
            int idx = 31 - 1;
            for (int i = 0; i < 31 - 1; ++i) {
                if (x < cam->filter_cdf[i]) { idx = i; break; }
            }
            float pdf = cam->filter_cdf[idx] - cam->filter_cdf[idx - 1];
            float u = cam->filter_bin_size*(idx + (x - cam->filter_cdf[idx - 1])/pdf);
            out[k] = negative ? -u : u;
        }
        fu = out[0]; fv = out[1];
    }
    v3 localD;
    if (thinlens) {
        /* ThinlensCamera::sampleDirection (ThinlensCamera.cpp:106-133); note: no + 0.5 on the pixel here */
        v3 planePos = V(-1.0f + ((float)px + fu)*2.0f*cam->pixel_size_x,
                        cam->ratio - ((float)py + fv)*2.0f*cam->pixel_size_x,
                        cam->plane_dist);
        planePos = vscale(planePos, cam->focus_dist/planePos.z);
        const float *m = cam->inv_xf;
        v3 lensPos = V(m[0]*lensP.x + m[1]*lensP.y + m[2]*lensP.z + m[3],
                       m[4]*lensP.x + m[5]*lensP.y + m[6]*lensP.z + m[7],
                       m[8]*lensP.x + m[9]*lensP.y + m[10]*lensP.z + m[11]);
        localD = vnorm(vsub(planePos, lensPos));
        if (cam->cat_eye > 0.0f) {
            float k = cam->cat_eye*cam->plane_dist;
            float dx = lensPos.x - k*localD.x/localD.z, dy = lensPos.y - k*localD.y/localD.z;
            if (dx*dx + dy*dy > sqr(cam->aperture_size))
                return vs(0.0f);                          /* sampleDirection fails: PathTracer.cpp:27-28 */
        }
    } else {
        localD = vnorm(V(-1.0f + ((float)px + 0.5f + fu)*2.0f*cam->pixel_size_x,
                         cam->ratio - ((float)py + 0.5f + fv)*2.0f*cam->pixel_size_x,
                         cam->plane_dist));
    }
    Ray ray;
    ray.o = lensP;
    ray.d = mat3_mul(cam->xf, localD);
    if (cam->type == TGHIP_CAMERA_EQUIRECTANGULAR) {
        /* EquirectangularCamera::sampleDirection / uvToDirection (cameras/EquirectangularCamera.cpp:26-36, 70-83): inv_xf holds _rot and 1 / res_y */
        float u = ((float)px + 0.5f + fu)*cam->pixel_size_x, v = ((float)py + 0.5f + fv)*cam->inv_xf[9];
        float phi = (u - 0.5f)*O_TWO_PI, theta = (1.0f - v)*O_PI;
        float sinTheta = sinf(theta);
        v3 l = V(cosf(phi)*sinTheta, -cosf(theta), sinf(phi)*sinTheta);
        const float *m = cam->inv_xf;
        ray.d = V(m[0]*l.x + m[1]*l.y + m[2]*l.z + 0.0f, m[3]*l.x + m[4]*l.y + m[5]*l.z + 0.0f, m[6]*l.x + m[7]*l.y + m[8]*l.z + 0.0f);
    }
    if (cam->type == TGHIP_CAMERA_CUBEMAP) {
        /* CubemapCamera::sampleDirection (cameras/CubemapCamera.cpp:153-166) with uvToFace (:95-106), uvToDirection (:107-112), faceToDirection (:74-80), the layout
         * tables of :10-46 and prepareForRender (:217-232); blade_count = the projection mode */
        static const int ResU[4] = {4, 3, 6, 1}, ResV[4] = {3, 4, 1, 6};
        static const int OffsetU[4][6] = {{2, 0, 1, 1, 1, 3}, {1, 1, 1, 1, 0, 2}, {0, 1, 2, 3, 4, 5}, {0, 0, 0, 0, 0, 0}};
        static const int OffsetV[4][6] = {{1, 1, 0, 2, 1, 1}, {1, 3, 0, 2, 1, 1}, {0, 0, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5}};
        static const int BasisU[4][6] = {{5, 4, 0, 0, 0, 1}, {5, 5, 5, 5, 0, 1}, {5, 4, 0, 0, 0, 1}, {5, 4, 0, 0, 0, 1}};
        static const int BasisV[4][6] = {{3, 3, 4, 5, 3, 3}, {3, 2, 0, 1, 3, 3}, {3, 3, 4, 5, 3, 3}, {3, 3, 4, 5, 3, 3}};
        static const float Basis[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
        const int mode = cam->blade_count;
        const float faceW = 1.0f/(float)ResU[mode], faceH = 1.0f/(float)ResV[mode];
        float u = ((float)px + 0.5f)*cam->pixel_size_x, v = ((float)py + 0.5f)*cam->inv_xf[9];
        int face = -1;
        for (int i = 0; i < 6 && face < 0; ++i) {
            float dx = u - (float)OffsetU[mode][i]*faceW, dy = v - (float)OffsetV[mode][i]*faceH;
            if (dx >= 0.0f && dy >= 0.0f && dx <= faceW && dy <= faceH) face = i;
        }
        if (face < 0)
            return vs(0.0f);                              /* sampleDirection fails: PathTracer.cpp:27-28 */
        u += fu*cam->pixel_size_x; v += fv*cam->inv_xf[9];
        float dx = u - (float)OffsetU[mode][face]*faceW, dy = v - (float)OffsetV[mode][face]*faceH;
        float ox = dx/faceW, oy = dy/faceH;
        const float *b = Basis[face], *bu = Basis[BasisU[mode][face]], *bv = Basis[BasisV[mode][face]];
        v3 l = vnorm(vadd(vadd(V(b[0], b[1], b[2]), vscale(V(bu[0], bu[1], bu[2]), ox*2.0f - 1.0f)), vscale(V(bv[0], bv[1], bv[2]), oy*2.0f - 1.0f)));
        const float *m = cam->inv_xf;
        ray.d = V(m[0]*l.x + m[1]*l.y + m[2]*l.z + 0.0f, m[3]*l.x + m[4]*l.y + m[5]*l.z + 0.0f, m[6]*l.x + m[7]*l.y + m[8]*l.z + 0.0f);
    }
    ray.tmin = 1e-4f; ray.tmax = INFINITY;            /* Ray ctor defaults, math/Ray.hpp:24 */

    v3 throughput = vs(1.0f), emission = vs(0.0f);
    Hit hit;
    Info info;
    int bounce = 0;
    int medium = s->num_media ? cam->medium : -1;      /* _scene->cam().medium() */
    int recorded = 0;                                   /* recordedOutputValues */
    float hitDistance = 0.0f;
    MediumState state = {1, 0};                         /* state.reset() */
    const int volumeNee = s->settings.enable_volume_light_sampling;
    c->closest_rays++;
    int didHit = scene_intersect(s, &ray, &hit, c->st);
    if (didHit) intersection_info(s, &ray, &hit, &info);
    int wasSpecular = 1;
    while ((didHit || medium >= 0) && bounce < maxBounces) {
        int hitSurface = 1;
        MediumSample ms;
        if (medium >= 0) {
            Ray seg = ray;
            seg.tmax = didHit ? hit.t : INFINITY;          /* ray.farT() after _scene->intersect */
            if (!medium_sampleDistance(s, medium, c->sampler, &seg, &state, &ms))
                return emission;
            throughput = vmul(throughput, ms.weight);      /* mediumSample.emission = 0 */
            hitSurface = ms.exited;
            if (hitSurface && !didHit)
                break;
        }
        if (hitSurface) {
        hitDistance += hit.t;                              /* PathTracer.cpp:64 */
        Local l = makeLocalScatterEvent(c, &info, &ray);

        /* TraceBase::handleSurface (TraceBase.cpp:516-568); terminate = it returned false */
        Event fe = make_event(c, &info, &l, TGHIP_LOBE_FORWARD);
        fe.wo = vneg(fe.wi);
        v3 transparency = bsdf_eval_rt(s, info.bsdf, &fe);
        float transparencyScalar = vavg(transparency);
        v3 wo = ray.d;
        v3 transmittance = vs(-1.0f);
        int terminate = 0;
        if (nextBoolean(c->sampler, transparencyScalar)) {
            throughput = vmul(throughput, vdivs(transparency, transparencyScalar));
        } else {
            if (nee && bounce < maxBounces - 1) {
                c->visRequest = c->aux ? &transmittance : NULL;
                emission = vadd(emission, vmul(estimateDirect(c, &info, &l, medium, bounce + 1), throughput));
                c->visRequest = NULL;
            }
            const TgHipObject *o = &s->objects[info.object];
            if (o->emission >= 0 && bounce >= minBounces) {
                if (!nee || wasSpecular || o->light < 0) {
                    LightHit lh; lh.u = info.u; lh.v = info.v; lh.backSide = info.backSide;
                    emission = vadd(emission, vmul(light_evalDirect(s, info.object, &lh), throughput));
                }
            }
            Event e = make_event(c, &info, &l, LOBE_ALL);
            if (!bsdf_sample_rt(s, info.bsdf, &e)) {
                terminate = 1;
            } else {
                wo = toGlobal(&l.frame, e.wo);
                if (!isConsistent(c, &info, &l, e.wo, wo)) {
                    terminate = 1;
                } else {
                    throughput = vmul(throughput, e.weight);
                    wasSpecular = (e.sampled & LOBE_SPECULAR) != 0;
                }
            }
        }
        if (c->aux && !recorded && (!wasSpecular || terminate)) {      /* PathTracer.cpp:78-96 */
            AuxSample *a = c->aux;
            a->has[TGHIP_AUX_DEPTH] = 1;  a->v[3] = hitDistance;
            a->has[TGHIP_AUX_NORMAL] = 1; a->v[4] = info.Ns.x; a->v[5] = info.Ns.y; a->v[6] = info.Ns.z;
            int ab = info.bsdf;                                         /* TransparencyBsdf: its base's albedo */
            if (s->bsdfs[ab].type == TGHIP_BSDF_TRANSPARENCY) ab = s->bsdfs[ab].sub0;
            v3 albedo = texture_eval(s, s->bsdfs[ab].albedo, info.u, info.v);
            if (s->objects[info.object].emission >= 0) {                /* isEmissive(): + evalDirect */
                LightHit lh; lh.u = info.u; lh.v = info.v; lh.backSide = info.backSide;
                albedo = vadd(albedo, light_evalDirect(s, info.object, &lh));
            }
            a->has[TGHIP_AUX_ALBEDO] = 1; a->v[7] = albedo.x; a->v[8] = albedo.y; a->v[9] = albedo.z;
            if (transmittance.x != -1.0f || transmittance.y != -1.0f || transmittance.z != -1.0f) {
                a->has[TGHIP_AUX_VISIBILITY] = 1; a->v[10] = vavg(transmittance);
            }
            recorded = 1;
        }
        if (terminate)
            return emission;
        medium = selectMedium(&s->objects[info.object], medium, vdot(wo, info.Ng) < 0.0f);   /* :561-563 */
        state.firstScatter = 1; state.bounce = 0;
        v3 hp = vadd(ray.o, vscale(ray.d, hit.t));      /* ray.hitpoint() */
        ray.o = hp; ray.d = wo; ray.tmin = info.epsilon; ray.tmax = INFINITY;
        } else {
            /* TraceBase::handleVolume (TraceBase.cpp:496-514) */
            wasSpecular = !volumeNee;
            if (volumeNee && bounce < maxBounces - 1)
                emission = vadd(emission, vmul(throughput, volumeEstimateDirect(c, &ms, medium, bounce + 1, ray.d)));
            v3 w; float pdf;
            phase_sample(&s->media[ms.medium], c->sampler, ray.d, &w, &pdf);
            ray.o = ms.p; ray.d = w; ray.tmin = 0.0f; ray.tmax = INFINITY;     /* ray.scatter(p, w, 0.0f); weight = 1 */
        }

        if (vmax3(throughput) == 0.0f)
            break;
        float roulettePdf = fmaxf(fabsf(throughput.x), fmaxf(fabsf(throughput.y), fabsf(throughput.z)));
        if (bounce > 2 && roulettePdf < 0.1f) {
            if (nextBoolean(c->sampler, roulettePdf))
                throughput = vdivs(throughput, roulettePdf);
            else
                return emission;
        }
        if (isnan(vsum(ray.d) + vsum(ray.o)))
            return vs(0.0f);
        if (isnan(vsum(throughput) + vsum(emission)))
            return vs(0.0f);

        bounce++;
        if (bounce < maxBounces) {
            c->closest_rays++;
            didHit = scene_intersect(s, &ray, &hit, c->st);
            if (didHit) intersection_info(s, &ray, &hit, &info);
        }
    }
    /* handleInfiniteLights (TraceBase.cpp:570-578, TraceableScene.hpp:194-209): the last infinite light wins */
    int objIdx = -1;                                   /* info.primitive after intersectInfinites, when it was asked */
    v3 envAlbedo = vs(0.0f);
    if (bounce >= minBounces && bounce < maxBounces && s->num_infinite_lights > 0) {
        for (uint32_t i = 0; i < s->num_infinite_lights; ++i) {      /* every infinite light is asked; the last hit stays in `data` */
            const TgHipObject *c = &s->objects[s->infinite_lights[i]];
            if (c->type != TGHIP_OBJ_INFINITE_SPHERE_CAP || vdot(ray.d, ld3(c->normal)) >= c->scale[0])
                objIdx = s->infinite_lights[i];
        }
        if (objIdx >= 0) {
            const TgHipObject *o = &s->objects[objIdx];
            float u = 0.0f, v = 0.0f;
            if (o->type == TGHIP_OBJ_INFINITE_SPHERE) inf_directionToUV(o, ray.d, &u, &v, NULL);
            envAlbedo = texture_eval(s, o->emission, u, v);              /* info.primitive->evalDirect(data, info) */
            if (!nee || wasSpecular || !(o->flags & TGHIP_OBJF_SAMPLE))
                emission = vadd(emission, vmul(throughput, envAlbedo));
        }
    }
    if (isnan(vsum(throughput) + vsum(emission)))
        return vs(0.0f);
    if (c->aux && !recorded) {                          /* PathTracer.cpp:133-140 */
        AuxSample *a = c->aux;
        if (bounce == 0) { a->has[TGHIP_AUX_DEPTH] = 1; a->v[3] = 0.0f; }
        a->has[TGHIP_AUX_NORMAL] = 1; a->v[4] = -ray.d.x; a->v[5] = -ray.d.y; a->v[6] = -ray.d.z;
        if (objIdx >= 0) { a->has[TGHIP_AUX_ALBEDO] = 1; a->v[7] = envAlbedo.x; a->v[8] = envAlbedo.y; a->v[9] = envAlbedo.z; }
    }
    return emission;
}

/* ---------------------------------------------------------------------------------------------
 * Exported entry points
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t samples, closest_rays, shadow_rays, nodes_visited, prims_tested;
} OracleCounters;

void oracle_trace_sample(const TgHipSceneDesc *s, uint32_t seed, uint32_t px, uint32_t py, uint32_t sampleIndex, float *rgb)
{
    Sampler smp;
    sampler_start(&smp, seed, px + py*(uint32_t)s->camera.res_x, sampleIndex);
    Ctx c = {s, &smp, NULL, 0, 0};
    v3 r = traceSample(&c, px, py);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

/* SampleRecord::addSample (path_tracer/SampleRecord.hpp:46-58) */
static inline void record_add(TgHipSampleRecord *r, v3 c)
{
    float x = c.x*0.2126f + c.y*0.7152f + c.z*0.0722f;   /* Vec3f::luminance (math/Vec.hpp:195-199) */
    r->sample_count++;
    float delta = x - r->mean;
    r->mean += delta/(float)r->sample_count;
    r->running_variance += delta*(x - r->mean);
}

/* one path under the Sobol' sampler of the tile with seed tileSeed (SobolPathSampler + PathTracer::traceSample) */
void oracle_trace_sample_sobol(const TgHipSceneDesc *s, uint32_t seed, uint32_t tileSeed, uint32_t px, uint32_t py, uint32_t sampleIndex, float *rgb)
{
    Sampler smp;
    sampler_start_sobol(&smp, s->sobol_matrices, tileSeed, seed, px + py*(uint32_t)s->camera.res_x, sampleIndex);
    Ctx c = {s, &smp, NULL, 0, 0};
    v3 r = traceSample(&c, px, py);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

/* debugging aid: one path with the Sobol' sampler, logging the numbers it draws; returns how many */
int oracle_trace_sample_log(const TgHipSceneDesc *s, uint32_t seed, uint32_t tileSeed, int sobol, uint32_t px, uint32_t py, uint32_t sampleIndex,
                            float *rgb, float *log, int log_cap)
{
    Sampler smp;
    uint32_t pixelIndex = px + py*(uint32_t)s->camera.res_x;
    if (sobol) sampler_start_sobol(&smp, s->sobol_matrices, tileSeed, seed, pixelIndex, sampleIndex);
    else       sampler_start(&smp, seed, pixelIndex, sampleIndex);
    smp.log = log; smp.log_cap = log_cap;
    Ctx c = {s, &smp, NULL, 0, 0};
    v3 r = traceSample(&c, px, py);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
    return smp.log_n;
}

/* One pass over the shard's tiles, renderTile semantics (PathTraceIntegrator.cpp:136-156) with the
 * framebuffer kept as sum + count (OutputBuffer.hpp:104-107 drops NaN/Inf samples without counting).
 * records (may be NULL): the SampleRecords, updated in the reference's order when TGHIP_PASS_RECORDS is set. */
/* OutputBuffer<T>::addSample (cameras/OutputBuffer.hpp:104-132) with _bufferB and _variance present, on the channels
 * [ch0, ch0 + nch) of one pixel; c = the sample's value */
static void aux_add(TgHipAuxPixel *px, int output, int ch0, int nch, const float *c)
{
    for (int k = 0; k < nch; ++k)
        if (isnan(c[k]) || isinf(c[k]))
            return;
    uint32_t sampleIdx = px->count[output]++;
    for (int k = 0; k < nch; ++k) {
        float *a = &px->a[ch0 + k], *b = &px->b[ch0 + k], *var = &px->variance[ch0 + k];
        float curr;
        if (sampleIdx > 0) {
            uint32_t sampleCountA = (sampleIdx + 1)/2, sampleCountB = sampleIdx/2;
            curr = (*a*(float)sampleCountA + *b*(float)sampleCountB)/(float)sampleIdx;
        } else {
            curr = *a;
        }
        float delta = c[k] - curr;
        curr += delta/(float)(sampleIdx + 1);
        *var += delta*(c[k] - curr);
        float *feature = (sampleIdx & 1) ? b : a;
        uint32_t perBufferSampleCount = sampleIdx/2 + 1;
        *feature += (c[k] - *feature)/(float)perBufferSampleCount;
    }
}

int oracle_render_aux(const TgHipSceneDesc *s, const TgHipPassDesc *pass, float *rgb_sum, uint32_t *count,
                      TgHipSampleRecord *records, TgHipAuxPixel *aux, OracleCounters *counters, int nthreads);

int oracle_render_records(const TgHipSceneDesc *s, const TgHipPassDesc *pass, float *rgb_sum, uint32_t *count,
                          TgHipSampleRecord *records, OracleCounters *counters, int nthreads)
{
    return oracle_render_aux(s, pass, rgb_sum, count, records, NULL, counters, nthreads);
}

/* aux (may be NULL; used when TGHIP_PASS_AUX is set): one TgHipAuxPixel per image pixel, updated per sample in the
 * reference's order (each pixel's samples in index order: renderTile, PathTraceIntegrator.cpp:136-156) */
int oracle_render_aux(const TgHipSceneDesc *s, const TgHipPassDesc *pass, float *rgb_sum, uint32_t *count,
                      TgHipSampleRecord *records, TgHipAuxPixel *aux, OracleCounters *counters, int nthreads)
{
    int w = s->camera.res_x, h = s->camera.res_y;
    int tilesX = (w + 15)/16, tilesY = (h + 15)/16;
    int varW = (w + 3)/4;
    uint32_t shardCount = pass->shard_count ? pass->shard_count : 1;
    const int sobol = (pass->flags & TGHIP_PASS_SOBOL) != 0;
    if (sobol && (!s->sobol_matrices || s->num_sobol_words < (uint64_t)TGHIP_SOBOL_DIMS*TGHIP_SOBOL_BITS || !pass->tile_seeds))
        return -1;
    if (!(pass->flags & TGHIP_PASS_RECORDS))
        records = NULL;
    if (!(pass->flags & TGHIP_PASS_AUX))
        aux = NULL;
    uint64_t tot[5] = {0, 0, 0, 0, 0};
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    #pragma omp parallel
    {
        uint64_t loc[5] = {0, 0, 0, 0, 0};
        TravStats st = {0, 0, 0};
        #pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < tilesX*tilesY; ++tile) {
            if (tghip_tile_owner((uint32_t)(tile % tilesX), (uint32_t)(tile/tilesX), shardCount) != pass->shard_index)
                continue;
            int x0 = (tile % tilesX)*16, y0 = (tile/tilesX)*16;
            for (int y = y0; y < y0 + 16 && y < h; ++y) {
                for (int x = x0; x < x0 + 16 && x < w; ++x) {
                    uint32_t pixelIndex = (uint32_t)(x + y*w);
                    uint32_t rec = (uint32_t)(x/4 + (y/4)*varW);
                    uint32_t begin = pass->spp_begin, end = pass->spp_end;
                    if (pass->record_count) {
                        begin = pass->record_index[rec];
                        end = begin + pass->record_count[rec];
                    }
                    for (uint32_t sidx = begin; sidx < end; ++sidx) {
                        Sampler smp;
                        if (sobol) sampler_start_sobol(&smp, s->sobol_matrices, pass->tile_seeds[tile], pass->seed, pixelIndex, sidx);
                        else       sampler_start(&smp, pass->seed, pixelIndex, sidx);
                        Ctx c = {s, &smp, counters ? &st : NULL, 0, 0};
                        AuxSample as;
                        memset(&as, 0, sizeof(as));
                        c.aux = aux ? &as : NULL;
                        v3 r = traceSample(&c, (uint32_t)x, (uint32_t)y);
                        loc[0]++; loc[1] += c.closest_rays; loc[2] += c.shadow_rays;
                        if (aux) {                       /* the addSample calls inside traceSample, then the colour (:148-152) */
                            TgHipAuxPixel *px = &aux[pixelIndex];
                            if (as.has[TGHIP_AUX_DEPTH]) aux_add(px, TGHIP_AUX_DEPTH, 3, 1, &as.v[3]);
                            if (as.has[TGHIP_AUX_NORMAL]) aux_add(px, TGHIP_AUX_NORMAL, 4, 3, &as.v[4]);
                            if (as.has[TGHIP_AUX_ALBEDO]) aux_add(px, TGHIP_AUX_ALBEDO, 7, 3, &as.v[7]);
                            if (as.has[TGHIP_AUX_VISIBILITY]) aux_add(px, TGHIP_AUX_VISIBILITY, 10, 1, &as.v[10]);
                            float col[3] = {r.x, r.y, r.z};
                            aux_add(px, TGHIP_AUX_COLOR, 0, 3, col);
                        }
                        if (records)
                            record_add(&records[rec], r);
                        if (isnan(r.x) || isnan(r.y) || isnan(r.z) || isinf(r.x) || isinf(r.y) || isinf(r.z))
                            continue;
                        rgb_sum[pixelIndex*3 + 0] += r.x;
                        rgb_sum[pixelIndex*3 + 1] += r.y;
                        rgb_sum[pixelIndex*3 + 2] += r.z;
                        count[pixelIndex]++;
                    }
                }
            }
        }
        loc[3] = st.nodes; loc[4] = st.prims;
        #pragma omp critical
        { for (int i = 0; i < 5; ++i) tot[i] += loc[i]; }
    }
    if (counters) {
        counters->samples += tot[0]; counters->closest_rays += tot[1]; counters->shadow_rays += tot[2];
        counters->nodes_visited += tot[3]; counters->prims_tested += tot[4];
    }
    return 0;
}

int oracle_render(const TgHipSceneDesc *s, const TgHipPassDesc *pass, float *rgb_sum, uint32_t *count,
                  OracleCounters *counters, int nthreads)
{
    return oracle_render_records(s, pass, rgb_sum, count, NULL, counters, nthreads);
}

/* ---------------------------------------------------------------------------------------------
 * The pass scheduler of PathTraceIntegrator (PathTraceIntegrator.cpp:27-134, 184-239) + SampleRecord.
 * ------------------------------------------------------------------------------------------- */
typedef struct {   /* path_tracer/SampleRecord.hpp:13-16 */
    uint32_t sampleCount, nextSampleCount, sampleIndex;
    float adaptiveWeight, mean, runningVariance;
} OracleRecord;

typedef struct { uint64_t state; } HostSampler;   /* UniformSampler with sequence 0 (UniformSampler.hpp:22-26) */
static uint32_t host_nextI(HostSampler *u)
{
    uint64_t oldState = u->state;
    u->state = oldState*6364136223846793005ULL + 1u;
    uint32_t xorShifted = (uint32_t)(((oldState >> 18u) ^ oldState) >> 27u);
    uint32_t rot = (uint32_t)(oldState >> 59u);
    return (xorShifted >> rot) | (xorShifted << ((uint32_t)(-(int32_t)rot) & 31));
}

static float record_errorEstimate(const OracleRecord *r)   /* SampleRecord.hpp:60-68 */
{
    float variance = r->runningVariance/(float)(r->sampleCount - 1u);
    float m2 = r->mean*r->mean;
    return variance/((float)r->sampleCount*(m2 > 1e-3f ? m2 : 1e-3f));
}

static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* PathTraceIntegrator::generateWork (:108-134) with errorPercentile95 (:44-60), dilateAdaptiveWeights (:62-88)
 * and distributeAdaptiveSamples (:90-112).  Returns 0 when there is nothing to render this pass. */
static int generate_work(OracleRecord *rec, int varW, int varH, int w, int h, HostSampler *smp,
                         uint32_t currentSpp, uint32_t nextSpp, int adaptive)
{
    int n = varW*varH;
    for (int i = 0; i < n; ++i)
        rec[i].sampleIndex += rec[i].nextSampleCount;
    int sppCount = (int)(nextSpp - currentSpp);
    if (adaptive && currentSpp >= 16u) {
        float *errors = (float *)malloc(sizeof(float)*(size_t)n);
        int ne = 0;
        for (int i = 0; i < n; ++i) {
            rec[i].adaptiveWeight = record_errorEstimate(&rec[i]);
            if (rec[i].adaptiveWeight > 0.0f)
                errors[ne++] = rec[i].adaptiveWeight;
        }
        float maxError = 0.0f;
        if (ne) {
            qsort(errors, (size_t)ne, sizeof(float), cmp_float);
            maxError = errors[((size_t)ne*95)/100];
        }
        free(errors);
        if (maxError == 0.0f)
            return 0;
        for (int i = 0; i < n; ++i)
            rec[i].adaptiveWeight = rec[i].adaptiveWeight < maxError ? rec[i].adaptiveWeight : maxError;
#define MAXW(a, b) ((a) > (b) ? (a) : (b))
        for (int y = 0; y < varH; ++y)
            for (int x = 0; x < varW; ++x) {
                int idx = x + y*varW;
                if (y < varH - 1) rec[idx].adaptiveWeight = MAXW(rec[idx].adaptiveWeight, rec[idx + varW].adaptiveWeight);
                if (x < varW - 1) rec[idx].adaptiveWeight = MAXW(rec[idx].adaptiveWeight, rec[idx + 1].adaptiveWeight);
            }
        for (int y = varH - 1; y >= 0; --y)
            for (int x = varW - 1; x >= 0; --x) {
                int idx = x + y*varW;
                if (y > 0) rec[idx].adaptiveWeight = MAXW(rec[idx].adaptiveWeight, rec[idx - varW].adaptiveWeight);
                if (x > 0) rec[idx].adaptiveWeight = MAXW(rec[idx].adaptiveWeight, rec[idx - 1].adaptiveWeight);
            }
#undef MAXW
        double totalWeight = 0.0;
        for (int i = 0; i < n; ++i)
            totalWeight += rec[i].adaptiveWeight;
        int adaptiveBudget = (sppCount - 1)*w*h;
        int budgetPerTile = adaptiveBudget/16;
        float weightToSampleFactor = (float)((double)budgetPerTile/totalWeight);
        float pixelPdf = 0.0f;
        for (int i = 0; i < n; ++i) {
            float fractionalSamples = rec[i].adaptiveWeight*weightToSampleFactor;
            int adaptiveSamples = (int)fractionalSamples;
            pixelPdf += fractionalSamples - (float)adaptiveSamples;
            if (normalizedUint(host_nextI(smp)) < pixelPdf) {
                adaptiveSamples++;
                pixelPdf -= 1.0f;
            }
            rec[i].nextSampleCount = (uint32_t)(adaptiveSamples + 1);
        }
    } else {
        for (int i = 0; i < n; ++i)
            rec[i].nextSampleCount = (uint32_t)sppCount;
    }
    return 1;
}

/* the scheduler alone, for pinning against the reference's dumps: state = the integrator's _sampler after dicing */
int oracle_generate_work(OracleRecord *rec, int w, int h, uint64_t *sampler_state, uint32_t currentSpp, uint32_t nextSpp, int adaptive)
{
    HostSampler smp = {*sampler_state};
    int r = generate_work(rec, (w + 3)/4, (h + 3)/4, w, h, &smp, currentSpp, nextSpp, adaptive);
    *sampler_state = smp.state;
    return r;
}

/* diceTiles (:27-42) after prepareForRender's `_sampler = UniformSampler(MathUtil::hash32(seed))` (:187): one
 * SobolPathSampler/UniformPathSampler seed per tile; returns the sampler state afterwards */
uint64_t oracle_dice_tiles(int w, int h, uint32_t seed, uint32_t *tile_seeds)
{
    HostSampler smp = {hash32(seed)};
    int tiles = ((w + 15)/16)*((h + 15)/16);
    for (int i = 0; i < tiles; ++i)
        tile_seeds[i] = hash32(host_nextI(&smp));
    return smp.state;
}

/* The whole render loop of the CLI (Shared.hpp:281-315: while (!done) { startRender; waitForCompletion; }).
 * records_out: max_passes x (varW*varH) records, dumped after every pass; returns the number of passes. */
int oracle_integrate_aux(const TgHipSceneDesc *s, uint32_t seed, uint32_t spp, uint32_t sppStep, int adaptive, int sobol,
                         float *rgb_sum, uint32_t *count, OracleRecord *records_out, int max_passes, uint32_t *pass_spp,
                         TgHipAuxPixel *aux, int nthreads);
int oracle_integrate(const TgHipSceneDesc *s, uint32_t seed, uint32_t spp, uint32_t sppStep, int adaptive, int sobol,
                     float *rgb_sum, uint32_t *count, OracleRecord *records_out, int max_passes, uint32_t *pass_spp, int nthreads)
{
    return oracle_integrate_aux(s, seed, spp, sppStep, adaptive, sobol, rgb_sum, count, records_out, max_passes, pass_spp, NULL, nthreads);
}

/* aux (may be NULL): the scene's output buffers (renderer.output_buffers), W x H TgHipAuxPixel, zero-initialised by the caller */
int oracle_integrate_aux(const TgHipSceneDesc *s, uint32_t seed, uint32_t spp, uint32_t sppStep, int adaptive, int sobol,
                         float *rgb_sum, uint32_t *count, OracleRecord *records_out, int max_passes, uint32_t *pass_spp,
                         TgHipAuxPixel *aux, int nthreads)
{
    int w = s->camera.res_x, h = s->camera.res_y;
    int varW = (w + 3)/4, varH = (h + 3)/4, n = varW*varH;
    int tiles = ((w + 15)/16)*((h + 15)/16);
    uint32_t *tileSeeds = (uint32_t *)malloc(sizeof(uint32_t)*(size_t)tiles);
    HostSampler smp = {oracle_dice_tiles(w, h, seed, tileSeeds)};
    OracleRecord *rec = (OracleRecord *)calloc((size_t)n, sizeof(OracleRecord));
    TgHipSampleRecord *dev = (TgHipSampleRecord *)calloc((size_t)n, sizeof(TgHipSampleRecord));
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t)*(size_t)n), *cnt = (uint32_t *)malloc(sizeof(uint32_t)*(size_t)n);
    uint32_t currentSpp = 0, nextSpp = sppStep < spp ? sppStep : spp;   /* Integrator::advanceSpp (Integrator.cpp:51-54) */
    int passes = 0, rc = 0;
    while (currentSpp < spp) {
        if (generate_work(rec, varW, varH, w, h, &smp, currentSpp, nextSpp, adaptive)) {
            for (int i = 0; i < n; ++i) { idx[i] = rec[i].sampleIndex; cnt[i] = rec[i].nextSampleCount; }
            TgHipPassDesc pass;
            memset(&pass, 0, sizeof(pass));
            pass.spp_begin = currentSpp; pass.spp_end = nextSpp; pass.seed = seed;
            pass.shard_index = 0; pass.shard_count = 1;
            pass.flags = TGHIP_PASS_RECORDS | (sobol ? TGHIP_PASS_SOBOL : 0u) | (aux ? TGHIP_PASS_AUX : 0u);
            pass.tile_seeds = tileSeeds; pass.record_index = idx; pass.record_count = cnt;
            rc = oracle_render_aux(s, &pass, rgb_sum, count, dev, aux, NULL, nthreads);
            if (rc) break;
            for (int i = 0; i < n; ++i) { rec[i].sampleCount = dev[i].sample_count; rec[i].mean = dev[i].mean; rec[i].runningVariance = dev[i].running_variance; }
        }
        currentSpp = nextSpp;
        nextSpp = currentSpp + sppStep < spp ? currentSpp + sppStep : spp;
        if (passes < max_passes) {
            if (records_out) memcpy(records_out + (size_t)passes*(size_t)n, rec, sizeof(OracleRecord)*(size_t)n);
            if (pass_spp) pass_spp[passes] = currentSpp;
        }
        passes++;
    }
    free(tileSeeds); free(rec); free(dev); free(idx); free(cnt);
    return rc ? rc : passes;
}

/* batched TraceableScene::intersect; also returns the exact visit counts that feed the
 * algorithmic-bytes model of the traversal kernel */
int oracle_trace_rays(const TgHipSceneDesc *s, const TgHipRay *rays, TgHipHit *hits, size_t n,
                      uint64_t *nodes_visited, uint64_t *prims_tested)
{
    TravStats st = {0, 0, 0};
    const int instWide = g_inst_wide;
    g_inst_wide = 0;                             /* (the device's ray query walks masters through the BVH2) */
    for (size_t i = 0; i < n; ++i) {
        Ray r = {ld3(rays[i].o), ld3(rays[i].d), rays[i].tmin, rays[i].tmax};
        Hit h;
        scene_intersect(s, &r, &h, &st);
        hits[i].t = h.t; hits[i].u = h.u; hits[i].v = h.v; hits[i].rec = h.rec;   /* the instance is not reported */
    }
    g_inst_wide = instWide;
    if (nodes_visited) *nodes_visited = st.nodes;
    if (prims_tested) *prims_tested = st.prims;
    return 0;
}

/* The DEVICE's shortcut for the walk of the reference's tree (csrc/hip/pt_kernels.h: flatClosestOrdered), restated here so that the CPU suite can
 * hold its claim against embree_top_walk above on rays chosen to tie (tests/test_flat_order.py): test every record against the ray's own tmax;
 * of the records hit whose leaf box the ray passes (a box missed under the ray's own tmax is never reached by the walk) keep the nearest hit b
 * and the second nearest distance t2; b is the walk's answer when it is the strict minimum and its box is not entered behind t2 -- whatever the
 * tree above the leaves looks like (a box contains its children's, so b's ancestors are passed and popped no later than b); with no such record
 * the walk finds nothing.  Otherwise the device walks the tree as the oracle does.
 * hits[i] = the answer, decided[i] = 1 when the shortcut applied.  Returns the number of rays on which shortcut and walk differ. */
static int top_leaf_box(const TgHipSceneDesc *s, int32_t rec, v3 *lo, v3 *hi)
{
    for (uint32_t n = 0; n < s->num_top_nodes; ++n)
        for (int i = 0; i < 4; ++i)
            if (s->top_nodes[n].child[i] == ~rec) { *lo = ld3(s->top_nodes[n].lower[i]); *hi = ld3(s->top_nodes[n].upper[i]); return 1; }
    return 0;
}
/* 1: the shortcut decides, *got = its answer (rec < 0: nothing hit); 0: the device walks the tree.
 * Round 5 (pt_kernels.h: flatClosestOrdered): no slab test inside the loop.  The loop keeps the nearest hit b (the first of equal ones), the
 * distance t2 of the second nearest hit whatever its box -- a lower bound of the second nearest hit the walk can reach: the rule only gets stricter
 * --, the number of records hit and whether a hit's distance is NaN; ONE slab test afterwards, of b's leaf box. */
static int flat_shortcut_decides(const TgHipSceneDesc *s, const Ray *rayIn, Hit *got)
{
    Ray ray = *rayIn;
    got->rec = -1; got->inst = -1; got->t = ray.tmax; got->u = got->v = 0.0f;
    float tb = INFINITY, t2 = INFINITY;
    uint32_t count = 0; int unordered = 0;
    for (uint32_t i = 0; i < s->num_recs; ++i) {
        Hit h; float tm = ray.tmax;
        h.rec = -1; h.inst = -1; h.t = tm; h.u = h.v = 0.0f;
        test_rec(s, i, &ray, &tm, &h, NULL, -1, -1);
        if (h.rec < 0) continue;
        if (h.t != h.t) unordered = 1;
        count++;
        if (h.t < tb) { t2 = tb; tb = h.t; *got = h; }
        else t2 = fminf(t2, h.t);
    }
    if (unordered) return 0;
    if (count == 0) return 1;
    v3 lo, hi; float entry;
    if (top_leaf_box(s, got->rec, &lo, &hi) && embree_box_near(&ray, lo, hi, &entry))
        return tb < t2 && entry <= t2;
    if (count == 1) { got->rec = -1; got->inst = -1; got->t = ray.tmax; got->u = got->v = 0.0f; return 1; }
    return 0;
}
/* Another formulation of the same idea (round 4's sketch; the device runs the rule above, which keeps ONE hit and decides a subset of what this
 * one decides): the loop keeps the THREE nearest hits, boxes or not; afterwards b = the first of them whose box the ray passes, and the second
 * nearest distance is bounded from below by the next stored hit, or by the third when more than three were hit (every hit not stored lies behind
 * it).  A lower bound only makes the rule stricter: decided rays stay right, a few more walk. */
static int flat_shortcut_decides_v2(const TgHipSceneDesc *s, const Ray *rayIn, Hit *got)
{
    Ray ray = *rayIn;
    Hit top[3]; int m = 0, count = 0;
    got->rec = -1; got->inst = -1; got->t = ray.tmax; got->u = got->v = 0.0f;
    for (uint32_t i = 0; i < s->num_recs; ++i) {
        Hit h; float tm = ray.tmax;
        h.rec = -1; h.inst = -1; h.t = tm; h.u = h.v = 0.0f;
        test_rec(s, i, &ray, &tm, &h, NULL, -1, -1);
        if (h.rec < 0) continue;
        if (h.t != h.t) return 0;                               /* a NaN distance (a ray IN a disk's plane: Disk::intersect divides 0 by 0 and accepts the result,
                                                                   as the reference does) cannot be ordered: the walk, where the disk's box usually culls it */
        count++;
        int k = m < 3 ? m : 3;                                  /* insert by t, equal distances keep their order */
        while (k > 0 && h.t < top[k - 1].t) { if (k < 3) top[k] = top[k - 1]; --k; }
        if (k < 3) { top[k] = h; if (m < 3) m++; }
    }
    int b = -1; float entryB = 0.0f;
    for (int k = 0; k < m && b < 0; ++k) {
        v3 lo, hi; float entry;
        if (top_leaf_box(s, top[k].rec, &lo, &hi) && embree_box_near(&ray, lo, hi, &entry)) { b = k; entryB = entry; }
    }
    if (b < 0) return count <= 3;                               /* nothing the walk can reach among the stored; unknown beyond them */
    const float t2lb = b + 1 < m ? top[b + 1].t : (count > m ? top[m - 1].t : INFINITY);
    *got = top[b];
    return top[b].t < t2lb && entryB <= t2lb;
}
size_t oracle_flat_device_form2(const TgHipSceneDesc *s, const TgHipRay *rays, uint8_t *decided, size_t n)
{
    size_t differing = 0;
    if (!s->top_nodes || !s->num_top_nodes) return (size_t)-1;
    for (size_t q = 0; q < n; ++q) {
        Ray ray = {ld3(rays[q].o), ld3(rays[q].d), rays[q].tmin, rays[q].tmax};
        Hit want, got;
        want.rec = -1; want.inst = -1; want.t = ray.tmax; want.u = want.v = 0.0f;
        embree_top_walk(s, &ray, &want, NULL, -1);
        const int dec = flat_shortcut_decides_v2(s, &ray, &got);
        if (decided) decided[q] = (uint8_t)dec;
        if (dec && (got.rec != want.rec || (got.rec >= 0 && (memcmp(&got.t, &want.t, 4) || memcmp(&got.u, &want.u, 4) || memcmp(&got.v, &want.v, 4))))) differing++;
    }
    return differing;
}
size_t oracle_flat_device_form(const TgHipSceneDesc *s, const TgHipRay *rays, TgHipHit *hits, uint8_t *decided, size_t n)
{
    size_t differing = 0;
    if (!s->top_nodes || !s->num_top_nodes) return (size_t)-1;
    for (size_t q = 0; q < n; ++q) {
        Ray ray = {ld3(rays[q].o), ld3(rays[q].d), rays[q].tmin, rays[q].tmax};
        Hit want, got;
        want.rec = -1; want.inst = -1; want.t = ray.tmax; want.u = want.v = 0.0f;
        embree_top_walk(s, &ray, &want, NULL, -1);
        const int dec = flat_shortcut_decides(s, &ray, &got);
        if (!dec) got = want;
        if (decided) decided[q] = (uint8_t)dec;
        if (hits) { hits[q].t = got.t; hits[q].u = got.u; hits[q].v = got.v; hits[q].rec = got.rec; }
        if (got.rec != want.rec || memcmp(&got.t, &want.t, 4) || memcmp(&got.u, &want.u, 4) || memcmp(&got.v, &want.v, 4)) differing++;
    }
    return differing;
}

/* transmittance kernels and distance samplers on their own (tests/test_media.py): k = 0 SS, 1 SM, 2 MS, 3 MM */
float oracle_trans_kernel(const TgHipMedium *m, int k, float tau) { return trans_kernel(m, k, tau); }
float oracle_trans_sigma_bar(const TgHipMedium *m) { return trans_sigmaBar(m); }
void oracle_trans_samples(const TgHipMedium *m, int startOnSurface, uint32_t seed, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        Sampler smp;
        sampler_start(&smp, seed, (uint32_t)i, 0);
        out[i] = trans_sample(m, &smp, startOnSurface);
    }
}

/* ---- unit-level hooks for the L1 parity tests (tests/test_oracle_units.py) ---------------- */
void oracle_rng_stream(uint32_t seed, uint32_t pixelIndex, uint32_t sampleIndex, int n, float *out)
{
    Sampler smp;
    sampler_start(&smp, seed, pixelIndex, sampleIndex);
    for (int i = 0; i < n; ++i) out[i] = next1D(&smp);
}

void oracle_camera_ray(const TgHipSceneDesc *s, uint32_t px, uint32_t py, float xi0, float xi1, float *o, float *d)
{
    /* re-uses traceSample's prologue through a replay sampler */
    const TgHipCamera *cam = &s->camera;
    float xi[2] = {xi0, xi1}, out[2] = {0, 0};
    if (cam->filter_type == TGHIP_FILTER_BOX) { out[0] = xi0 - 0.5f; out[1] = xi1 - 0.5f; }
    else if (cam->filter_type == TGHIP_FILTER_TABULATED) {
        for (int k = 0; k < 2; ++k) {
            float x = xi[k];
            int negative = x < 0.5f;
            x = negative ? x*2.0f : (x - 0.5f)*2.0f;
            int idx = 30;
            for (int i = 0; i < 30; ++i) if (x < cam->filter_cdf[i]) { idx = i; break; }
            float pdf = cam->filter_cdf[idx] - cam->filter_cdf[idx - 1];
            float u = cam->filter_bin_size*(idx + (x - cam->filter_cdf[idx - 1])/pdf);
            out[k] = negative ? -u : u;
        }
    }
    v3 localD = vnorm(V(-1.0f + ((float)px + 0.5f + out[0])*2.0f*cam->pixel_size_x,
                        cam->ratio - ((float)py + 0.5f + out[1])*2.0f*cam->pixel_size_x, cam->plane_dist));
    v3 dd = mat3_mul(cam->xf, localD);
    o[0] = cam->pos[0]; o[1] = cam->pos[1]; o[2] = cam->pos[2];
    d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
}

/* eval + pdf of bsdf `bi` for local directions wi, wo at texture coordinate uv, radiance transport */
void oracle_bsdf_eval(const TgHipSceneDesc *s, int bi, const float *wi, const float *wo, const float *uv,
                      uint32_t requested, float *f, float *pdf)
{
    Event e;
    memset(&e, 0, sizeof(e));
    e.wi = ld3(wi); e.wo = ld3(wo); e.weight = vs(1.0f); e.pdf = 1.0f;
    e.requested = requested; e.u = uv[0]; e.v = uv[1];
    v3 r = bsdf_eval_rt(s, bi, &e);
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
    *pdf = bsdf_pdf(s, bi, &e);
}

/* sample bsdf `bi` with the given uniform numbers; returns 0 if sampling failed */
int oracle_bsdf_sample(const TgHipSceneDesc *s, int bi, const float *wi, const float *uv, uint32_t requested,
                       const float *xi, int nxi, float *wo, float *weight, float *pdf, uint32_t *sampledLobe, int *consumed)
{
    Sampler smp;
    memset(&smp, 0, sizeof(smp));
    smp.replay = xi; smp.replay_n = nxi;
    Event e;
    memset(&e, 0, sizeof(e));
    e.wi = ld3(wi); e.weight = vs(1.0f); e.pdf = 1.0f;
    e.requested = requested; e.u = uv[0]; e.v = uv[1]; e.sampler = &smp;
    int ok = bsdf_sample_rt(s, bi, &e);
    wo[0] = e.wo.x; wo[1] = e.wo.y; wo[2] = e.wo.z;
    weight[0] = e.weight.x; weight[1] = e.weight.y; weight[2] = e.weight.z;
    *pdf = e.pdf; *sampledLobe = e.sampled;
    if (consumed) *consumed = smp.replay_pos;
    return ok;
}

/* light sampleDirect for light slot `li` from point p; returns 0 when the sample is rejected */
int oracle_light_sample(const TgHipSceneDesc *s, int li, const float *p, float xi0, float xi1, float *d, float *dist, float *pdf)
{
    float xi[2] = {xi0, xi1};
    Sampler smp;
    memset(&smp, 0, sizeof(smp));
    smp.replay = xi; smp.replay_n = 2;
    v3 dd = vs(0.0f);
    int ok = light_sampleDirect(s, s->lights[li], ld3(p), &smp, &dd, dist, pdf);
    d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
    return ok;
}

void oracle_texture_eval(const TgHipSceneDesc *s, int tex, float u, float v, float *rgb)
{
    v3 r = texture_eval(s, tex, u, v);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}
