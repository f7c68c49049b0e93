// Device-side scene access, textures, BSDFs and lights for the path_tracer_hip kernels.
//
// What the reference does through virtual calls on Texture/Bsdf/Primitive objects
// (src/core/textures, src/core/bsdfs, src/core/primitives) is a tagged-union interpreter over the
// flattened tables of include/tungsten_hip.h here.  Nested BSDFs (smooth_coat -> substrate,
// mixed, transparency) are unrolled at compile time by a depth template instead of recursion.
#ifndef TGAMD_PT_SCENE_H_
#define TGAMD_PT_SCENE_H_

#include "pt_math.h"
#include "pt_erfinv_table.h"
#include "../../../include/tungsten_hip.h"

struct DeviceScene {
    const float4 * __restrict__ nodes;        // 4 x float4 per TgHipBvhNode
    const float4 * __restrict__ recs;         // 3 x float4 per TgHipPrimRec
    const float4 * __restrict__ wide;         // 5 x float4 per TgHipWideNode (nullptr: the scene has no wide BVH); ONE allocation with recs:
    uint32_t recs_offset;                     //   recs == (char *)wide + recs_offset, so a lane addresses either with one 32-bit offset
    uint32_t wide_stride;                     // bytes between wide nodes: 80, or 128 ("wide_node_stride" option: one node per cache line)
    const float4 * __restrict__ tri_attrs;    // 4 x float4 per TgHipTriAttr
    const TgHipObject * __restrict__ objects;
    const int32_t * __restrict__ lights;
    const int32_t * __restrict__ infinite_lights;
    const TgHipBsdf * __restrict__ bsdfs;
    const TgHipTexture * __restrict__ textures;
    const float * __restrict__ texels;
    const float * __restrict__ dist;
    const uint8_t * __restrict__ rec_class;    // shading class of each primitive record's bsdf (built at upload, DESIGN.md "Kernels")
    const float * __restrict__ light_tris;     // sampled mesh emitters: cdf + triangles (include/tungsten_hip.h)
    const uint16_t * __restrict__ guide;       // CDF guide tables of the samplable bitmaps (built at upload, bitmapSample)
    const int32_t * __restrict__ tex_guide;    // per texture: offset of its tables in `guide`, -1 = none
    // Built at upload for the envmap-sampling chain (bitmapSample): per samplable bitmap the conditional (row) tables as
    // interleaved (cdf[i], pdf[i]) pairs, (w + 1) per row, so that one 32-byte window holds what a CDF inversion needs;
    // tex_rows[tex] = the bitmap's first pair (-1: none).  env_*: the MARGINAL tables (mpdf[h] mcdf[h + 1], guide[513]) of the
    // texture env_tex -- the scene's first sampled environment map --, which the shading kernels copy into LDS when they fit.
    const float2 * __restrict__ rows;
    const int32_t * __restrict__ tex_rows;
    int32_t env_tex, env_h;
    // The scene's ONE quad, in a scene of triangles otherwise (a ground plane under meshes: the metric's scene), or -1.  The decoupled walks test
    // it once per ray before the walk and skip it inside (its bit in the `reserved` word of the wide node that holds it, set by the shim):
    // nearly every turn of a wave used to run the quad test -- a dependent load of the quad's normal, a division -- for the one or two lanes
    // that happened to be at that record.  Same hit: a quad accepts t <= tmax and a triangle t < tmax, so the quad wins a tie whichever is
    // tested first; with two quads the order between THEM would matter, hence exactly one (tungsten_hip.hip: hoistedRec).
    int32_t hoisted_rec;
    const float *env_marginal;
    const uint16_t *env_guide;
    const uint32_t * __restrict__ sobol;       // Sobol' generator matrices (nullptr unless the scene carries them)
    uint32_t num_nodes, num_recs, num_objects, num_lights, num_infinite_lights, num_bsdfs, num_textures;
    uint32_t num_instances;                    // instance records (0: single-level scene, A_EMI.w carries nothing)
    const uint32_t * __restrict__ inst_prims;  // leaf slots of the reference's instance trees -> instance record (TgHipSceneDesc::inst_prims)
    const float4 * __restrict__ inst_leaf_boxes;   // 2 x float4 per leaf slot: the leaf's box as its parent holds it (lo, hi), at the leaf's first slot
    const float4 * __restrict__ inst_tight_boxes;  // 2 x float4 per top-level record: an instance record's tight world-space box (lo, hi)
    const float * __restrict__ top_nodes;      // flat lists of analytic primitives: the reference's top-level Embree tree, 28 floats per TgHipTopNode
                                               //   (nullptr: the scene carries none and walks the plain list / the BVH) -- pt_kernels.h: flatClosestOrdered
    const float4 * __restrict__ flat_boxes;    // ... and per record the box of its leaf in that tree (lo, hi)
    const TgHipMedium * __restrict__ media;    // participating media (nullptr / 0: none)
    uint32_t num_media;
    const TgHipCamera * __restrict__ camera;   // in device memory (56 dwords: read through the scalar cache where it is used, not held in SGPRs)
    TgHipSettings settings;
};

#define PT_MAX_BSDF_DEPTH 3

#define LOBE_ALL              (TGHIP_LOBE_GLOSSY_R | TGHIP_LOBE_GLOSSY_T | TGHIP_LOBE_DIFFUSE_R | TGHIP_LOBE_DIFFUSE_T | \
                               TGHIP_LOBE_SPECULAR_R | TGHIP_LOBE_SPECULAR_T | TGHIP_LOBE_ANISOTROPIC)
#define LOBE_SPECULAR         (TGHIP_LOBE_SPECULAR_R | TGHIP_LOBE_SPECULAR_T)
#define LOBE_TRANSMISSIVE     (TGHIP_LOBE_GLOSSY_T | TGHIP_LOBE_DIFFUSE_T | TGHIP_LOBE_SPECULAR_T)
#define LOBE_ALL_BUT_SPECULAR (~(uint32_t)(LOBE_SPECULAR | TGHIP_LOBE_FORWARD))

// M is the compile-time set of BSDF types (bit = 1 << TGHIP_BSDF_*) a kernel variant has to handle: cases
// outside M fold away, which is what keeps the Lambert-only shading kernel small (DESIGN.md "Kernels").
#define BSDF_BIT(t) (1u << (t))
#define BSDF_MASK_ALL 0xFFFFFFFFu
// The upper bits of M say which scene FEATURES a shading-kernel variant has to handle (all set in BSDF_MASK_ALL);
// code for absent features folds away like BSDF types do.
#define FEAT_BITMAP     (1u << 24)   /* bitmap textures (incl. environment maps)                      */
#define FEAT_INFINITE   (1u << 25)   /* infinite-sphere emitters                                       */
#define FEAT_MULTILIGHT (1u << 26)   /* more than one sampled light (TraceBase::chooseLight's pdf loop) */
#define FEAT_TRIANGLES  (1u << 27)   /* triangle records (attribute gather, smooth normals)            */
#define FEAT_SOLIDS     (1u << 28)   /* sphere records; sphere / cube emitters as sampled lights       */
#define FEAT_ALL        (FEAT_BITMAP | FEAT_INFINITE | FEAT_MULTILIGHT | FEAT_TRIANGLES | FEAT_SOLIDS)
#define FEAT_MESHLIGHT  (1u << 29)   /* triangle-mesh emitters as sampled lights: only in the MASK_FULL / BSDF_MASK_ALL variants */
#define FEAT_QMC        (1u << 30)   /* TGHIP_PASS_SOBOL / TGHIP_PASS_RECORDS passes: every variant has a twin with this bit (launchShade) */
#define FEAT_INSTANCES  (1u << 31)   /* hits reached through an instance record (primitives/Instance.cpp): only in MASK_FULL / BSDF_MASK_ALL */
#define FEAT_MEDIA      (1u << 23)   /* participating media (media/HomogeneousMedium.cpp): only in the BSDF_MASK_ALL variant */
#define FEAT_AUX        (1u << 22)   /* TGHIP_PASS_AUX passes (auxiliary output buffers): only in the BSDF_MASK_ALL variant */
#define FEAT_CYLINDER   (1u << 21)   /* cylinder primitives / emitters (primitives/Cylinder.cpp): only in the BSDF_MASK_ALL variant */
#define FEAT_PHONG      (1u << 19)   /* microfacet BSDFs with the Phong distribution (pow(double, double)): only in the MASK_FULL / BSDF_MASK_ALL variants */
#define FEAT_BUMP       (1u << 20)   /* bump-mapped shading frames (Primitive::setupTangentFrame, TgHipBsdf::bump1): only in the BSDF_MASK_ALL variant */
#define MASK_FULL       (BSDF_MASK_ALL & ~(FEAT_QMC | FEAT_MEDIA | FEAT_AUX | FEAT_CYLINDER | FEAT_BUMP))
// next1D of the path's sampler inside code templated on M
#define RNG1D(r) rngNext1DT<(M & FEAT_QMC) != 0>(r)

// ---------------------------------------------------------------------------------------------
// Textures (ConstantTexture, CheckerTexture.cpp:64-69, BitmapTexture.cpp:298-352)
// ---------------------------------------------------------------------------------------------
// One load whatever the format: three floats from the texel's address, a one-channel texture's value taken from the first (the shim leaves
// 16 readable bytes behind the texel array).  With a branch per format around the load, the four texels of a bilinear lookup were four
// SEQUENTIAL memory round trips -- each load inside its own divergent region, each followed by s_waitcnt vmcnt(0) -- in every shading variant
// with bitmap textures; now the four loads are issued back to back (PT_TEXEL_BRANCH = 1 gives the old form back for the A/B).
#ifndef PT_TEXEL_BRANCH
#define PT_TEXEL_BRANCH 0
#endif
#ifndef PT_TEXEL_STAGES
#define PT_TEXEL_STAGES 2
#endif
PT_DEV f3 bitmapTexel(const DeviceScene &s, const TgHipTexture &t, int x, int y)
{
    const float *tex = s.texels + t.texel_offset;
#if PT_TEXEL_BRANCH
    if (t.flags & TGHIP_TEXF_RGB) {
        const float *p = tex + ((size_t)x + (size_t)y*t.w)*3;
        return mk3(p[0], p[1], p[2]);
    }
    return splat3(tex[(size_t)x + (size_t)y*t.w]);
#else
    const bool rgb = (t.flags & TGHIP_TEXF_RGB) != 0;
    const float *p = tex + ((size_t)x + (size_t)y*t.w)*(rgb ? 3u : 1u);
    const float a = p[0], b = p[1], c = p[2];
    return mk3(a, rgb ? b : a, rgb ? c : a);
#endif
}

// i mod n for n > 0 and any i, result in [0, n): float-reciprocal quotient + fix-up instead of integer division
PT_DEV int wrapIndex(int i, int n)
{
    int q = (int)floorf((float)i/(float)n);
    int r = i - q*n;
    if (r < 0) r += n;
    if (r >= n) r -= n;
    return r;
}

template<uint32_t M>
PT_DEV f3 textureEval(const DeviceScene &s, int texIdx, float u0, float v0)
{
    const TgHipTexture &t = s.textures[texIdx];
    if (t.type == TGHIP_TEX_CONSTANT)
        return ld3(t.value);
    if (t.type == TGHIP_TEX_CHECKER) {
        int ui = (int)(u0*(float)t.res_u), vi = (int)(v0*(float)t.res_v);
        return ((ui ^ vi) & 1) ? ld3(t.on_color) : ld3(t.off_color);
    }
    if (!(M & FEAT_BITMAP))
        return splat3(0.0f);
    int w = t.w, h = t.h;
    float u = u0*w;
    float v = (1.0f - v0)*h;
    bool linear = (t.flags & TGHIP_TEXF_LINEAR) && (t.flags & TGHIP_TEXF_VALID);
    if (linear) { u -= 0.5f; v -= 0.5f; }
    int iu0 = u < 0.0f ? -(int)(-u) - 1 : (int)u;
    int iv0 = v < 0.0f ? -(int)(-v) - 1 : (int)v;
    int iu1 = iu0 + 1, iv1 = iv0 + 1;
    u -= iu0; v -= iv0;
    if (!(t.flags & TGHIP_TEXF_CLAMP)) {
        iu0 = wrapIndex(iu0, w); iu1 = wrapIndex(iu1, w);
        iv0 = wrapIndex(iv0, h); iv1 = wrapIndex(iv1, h);
    } else {
        iu0 = min(max(iu0, 0), w - 1); iu1 = min(max(iu1, 0), w - 1);
        iv0 = min(max(iv0, 0), h - 1); iv1 = min(max(iv1, 0), h - 1);
    }
#if PT_TEXEL_BRANCH
    if (!linear)
        return bitmapTexel(s, t, iu0, iv0);
#else
    // (a nearest-neighbour texture fetches its one texel four times -- one cache line -- instead of branching around three of the loads)
    if (!linear) { iu1 = iu0; iv1 = iv0; }
#endif
#if PT_TEXEL_BRANCH
    f3 x00 = bitmapTexel(s, t, iu0, iv0), x01 = bitmapTexel(s, t, iu1, iv0);
    f3 x10 = bitmapTexel(s, t, iu0, iv1), x11 = bitmapTexel(s, t, iu1, iv1);
#else
    f3 x00, x01, x10, x11;
    {
        // the four texels as four 12-byte loads issued back to back and ONE wait: the empty asm below consumes all twelve words, so every load
        // is issued before it and nothing that uses a texel is scheduled between the loads (left alone, the scheduler puts the format selects of
        // each texel right behind its load -- a wait per load, four memory round trips per lookup)
        const float *tex = s.texels + t.texel_offset;
        const bool rgb = (t.flags & TGHIP_TEXF_RGB) != 0;
        const uint32_t st = rgb ? 3u : 1u;
        const float *p00 = tex + ((size_t)iu0 + (size_t)iv0*t.w)*st, *p01 = tex + ((size_t)iu1 + (size_t)iv0*t.w)*st;
        const float *p10 = tex + ((size_t)iu0 + (size_t)iv1*t.w)*st, *p11 = tex + ((size_t)iu1 + (size_t)iv1*t.w)*st;
#if PT_TEXEL_STAGES == 2
        // (two stages: the two ROWS first -- two cache lines, two misses in flight --, then each row's neighbour, which is in the line that just
        // arrived; all four at once made the headline 1.3 % slower: the neighbour's request goes to L2 again while its line is still on its way)
        float a0 = p00[0], b0 = p00[1], c0 = p00[2], a2 = p10[0], b2 = p10[1], c2 = p10[2];
        asm volatile("" : "+v"(a0), "+v"(b0), "+v"(c0), "+v"(a2), "+v"(b2), "+v"(c2));
        float a1 = p01[0], b1 = p01[1], c1 = p01[2], a3 = p11[0], b3 = p11[1], c3 = p11[2];
        asm volatile("" : "+v"(a1), "+v"(b1), "+v"(c1), "+v"(a3), "+v"(b3), "+v"(c3));
#else
        float a0 = p00[0], b0 = p00[1], c0 = p00[2], a1 = p01[0], b1 = p01[1], c1 = p01[2];
        float a2 = p10[0], b2 = p10[1], c2 = p10[2], a3 = p11[0], b3 = p11[1], c3 = p11[2];
        asm volatile("" : "+v"(a0), "+v"(b0), "+v"(c0), "+v"(a1), "+v"(b1), "+v"(c1), "+v"(a2), "+v"(b2), "+v"(c2), "+v"(a3), "+v"(b3), "+v"(c3));
#endif
        x00 = mk3(a0, rgb ? b0 : a0, rgb ? c0 : a0); x01 = mk3(a1, rgb ? b1 : a1, rgb ? c1 : a1);
        x10 = mk3(a2, rgb ? b2 : a2, rgb ? c2 : a2); x11 = mk3(a3, rgb ? b3 : a3, rgb ? c3 : a3);
    }
#endif
    f3 r = (x00*(1.0f - u) + x01*u)*(1.0f - v) + (x10*(1.0f - u) + x11*u)*v;
#if !PT_TEXEL_BRANCH
    // (a select, not a branch: the interpolation is computed either way, so that the compiler cannot sink three of the four loads into it)
    const f3 rs = r*t.scale;
    return mk3(linear ? rs.x : x00.x, linear ? rs.y : x00.y, linear ? rs.z : x00.z);
#else
    return r*t.scale;
#endif
}

// Distribution2D::warp / pdf (sampling/Distribution2D.hpp:68-83) on the flattened tables
PT_DEV int upperBoundIdx(const float *a, int n, float x)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
// what a CDF inversion already fetched for the texel it picked: bitmapPdf of the same texel needs no further loads
struct BitmapPick { int row, column; float pdfRC, mpdfR; };
PT_DEV float bitmapPdf(const DeviceScene &s, int texIdx, const TgHipTexture &t, float u, float v, const BitmapPick *pick = nullptr)   /* BitmapTexture.cpp:452-455 */
{
    int row = (int)((1.0f - v)*t.h), column = (int)(u*t.w);
    row = min(max(row, 0), t.h - 1);
    column = min(max(column, 0), t.w - 1);
    if (pick && pick->row == row && pick->column == column)
        return pick->pdfRC*pick->mpdfR*t.w*t.h;
    // (two reads, not one through a selected pointer: the environment map's marginal table is in LDS, the others' in global memory)
    const float mpdfR = texIdx == s.env_tex ? s.env_marginal[row] : (s.dist + t.dist_offset)[row];
    const int ro = s.tex_rows[texIdx];
    if (ro >= 0)
        return at32(s.rows, (uint32_t)ro + (uint32_t)row*(uint32_t)(t.w + 1) + (uint32_t)column).y*mpdfR*t.w*t.h;
    const float *pdf = s.dist + t.dist_offset + t.h + t.h + 1;
    return pdf[(size_t)row*t.w + column]*mpdfR*t.w*t.h;
}
// Guide tables (built by the shim at upload) make the two CDF inversions of Distribution2D::warp short dependent
// chains instead of 9- and 10-step binary searches over L2-resident arrays: for a CDF a[0..n] and B buckets,
// g[b] = upper_bound(a, b/B), so for x in [b/B, (b+1)/B) the answer lies in [g[b], g[b+1]].  B is a power of two
// (x*B is exact), and the final search inside the window is the same upper_bound, so the result is identical.
#define PT_GUIDE_MARGINAL 512
#define PT_GUIDE_ROW      256
PT_DEV int upperBoundGuided(const float *a, const uint16_t *g, int buckets, float x)
{
    int b = min((int)(x*(float)buckets), buckets - 1);
    int lo = g[b], hi = g[b + 1];
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
// The same inversion on a row of (cdf, pdf) pairs: the guide's window [lo, hi] usually spans a few entries, so the three
// pairs from lo - 1 on (loads issued together) decide it without a dependent search loop; wider windows fall
// back to the loop.  Returns upper_bound's index and the pair at index - 1 (the texel picked).
PT_DEV int upperBoundGuidedPairs(const float2 *a, const uint16_t *g, int buckets, float x, int n, float2 &picked)
{
    int b = min((int)(x*(float)buckets), buckets - 1);
    int lo = g[b], hi = g[b + 1];
    // lo >= 1 (a[0] = 0 <= x) and the pairs lo - 1 .. lo + 1 exist when lo + 1 <= n
    if (hi - lo <= 2 && lo >= 1 && lo + 1 <= n) {
        const float2 p0 = a[lo - 1], p1 = a[lo], p2 = a[lo + 1];
        // upper_bound over [lo, hi]: the first index whose cdf exceeds x (the binary search's answer: cdfs are non-decreasing)
        int r = hi;
        if (hi > lo + 1 && !(p2.x <= x)) r = lo + 1;
        if (hi > lo && !(p1.x <= x)) r = lo;
        picked = r == lo ? p0 : r == lo + 1 ? p1 : p2;
        return r;
    }
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid].x <= x) lo = mid + 1; else hi = mid; }
    picked = a[lo - 1];
    return lo;
}
PT_DEV void bitmapSample(const DeviceScene &s, int texIdx, const TgHipTexture &t, float xi0, float xi1, float &u, float &v, BitmapPick &pick)   /* :433-439 */
{
    const bool env = texIdx == s.env_tex;
    const int go = s.tex_guide[texIdx];
    const int ro = s.tex_rows[texIdx];
    int row;
    float mpdfR, mcdfR;
    // (the sampled environment map's marginal tables are in LDS, every other texture's in global memory: two copies of the search, so that
    // neither reads through a pointer of unknown address space -- stageSceneTables)
    if (env) {
        const float *mpdf = s.env_marginal, *mcdf = mpdf + t.h;
        row = upperBoundGuided(mcdf, s.env_guide, PT_GUIDE_MARGINAL, xi1) - 1;     // (env_tex >= 0 only for textures with guide tables)
        mpdfR = mpdf[row]; mcdfR = mcdf[row];
    } else {
        const float *mpdf = s.dist + t.dist_offset, *mcdf = mpdf + t.h;
        if (go >= 0) row = upperBoundGuided(mcdf, s.guide + go, PT_GUIDE_MARGINAL, xi1) - 1;
        else         row = upperBoundIdx(mcdf, t.h + 1, xi1) - 1;
        mpdfR = mpdf[row]; mcdfR = mcdf[row];
    }
    float nv = clampf((xi1 - mcdfR)/mpdfR, 0.0f, 1.0f);
    int column;
    float cdfC, pdfC;
    if (go >= 0 && ro >= 0) {
        const float2 *rowStart = &at32(s.rows, (uint32_t)ro + (uint32_t)row*(uint32_t)(t.w + 1));
        float2 picked;
        column = upperBoundGuidedPairs(rowStart, s.guide + go + (PT_GUIDE_MARGINAL + 1) + row*(PT_GUIDE_ROW + 1), PT_GUIDE_ROW, xi0, t.w, picked) - 1;
        cdfC = picked.x; pdfC = picked.y;
    } else {
        const float *pdf = s.dist + t.dist_offset + t.h + t.h + 1;
        const float *rowStart = pdf + (size_t)t.w*t.h + (size_t)row*(t.w + 1);
        if (go >= 0) column = upperBoundGuided(rowStart, s.guide + go + (PT_GUIDE_MARGINAL + 1) + row*(PT_GUIDE_ROW + 1), PT_GUIDE_ROW, xi0) - 1;
        else         column = upperBoundIdx(rowStart, t.w + 1, xi0) - 1;
        cdfC = rowStart[column]; pdfC = pdf[(size_t)row*t.w + column];
    }
    float nu = clampf((xi0 - cdfC)/pdfC, 0.0f, 1.0f);
    u = (nu + column)/t.w;
    v = 1.0f - (nv + row)/t.h;
    pick.row = row; pick.column = column; pick.pdfRC = pdfC; pick.mpdfR = mpdfR;
}

// ---------------------------------------------------------------------------------------------
// Fresnel + microfacet (bsdfs/Fresnel.hpp:75-138, bsdfs/Microfacet.hpp:27-130)
// ---------------------------------------------------------------------------------------------
PT_DEV float dielectricReflectance(float eta, float cosThetaI, float &cosThetaT)
{
    if (cosThetaI < 0.0f) {
        eta = 1.0f/eta;
        cosThetaI = -cosThetaI;
    }
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f) {
        cosThetaT = 0.0f;
        return 1.0f;
    }
    cosThetaT = sqrtf(fmaxf(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT);
    float Rp = (eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI);
    return (Rs*Rs + Rp*Rp)*0.5f;
}
// Fresnel::thinFilmReflectance / thinFilmReflectanceInterference (Fresnel.hpp:15-28, 39-67; `thickness` in nanometres)
PT_DEV float thinFilmReflectance(float eta, float cosThetaI, float &cosThetaT)
{
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f) {
        cosThetaT = 0.0f;
        return 1.0f;
    }
    cosThetaT = sqrtf(fmaxf(1.0f - sinThetaTSq, 0.0f));
    float Rs = sqr((eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT));
    float Rp = sqr((eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI));
    return 1.0f - ((1.0f - Rs)/(1.0f + Rs) + (1.0f - Rp)/(1.0f + Rp))*0.5f;
}
PT_DEV f3 thinFilmReflectanceInterference(float eta, float cosThetaI, float thickness, float &cosThetaT)
{
    const f3 invLambdas = mk3(1.0f/650.0f, 1.0f/510.0f, 1.0f/475.0f);
    float cosThetaISq = cosThetaI*cosThetaI;
    float sinThetaISq = 1.0f - cosThetaISq;
    float invEta = 1.0f/eta;
    float sinThetaTSq = eta*eta*sinThetaISq;
    if (sinThetaTSq > 1.0f) {
        cosThetaT = 0.0f;
        return splat3(1.0f);
    }
    cosThetaT = sqrtf(1.0f - sinThetaTSq);
    float Ts = 4.0f*eta*cosThetaI*cosThetaT/sqr(eta*cosThetaI + cosThetaT);
    float Tp = 4.0f*eta*cosThetaI*cosThetaT/sqr(eta*cosThetaT + cosThetaI);
    float Rs = 1.0f - Ts;
    float Rp = 1.0f - Tp;
    f3 phi = invLambdas*(thickness*cosThetaT*PT_FOUR_PI*invEta);
    f3 cosPhi = mk3(cosfH(phi.x), cosfH(phi.y), cosfH(phi.z));
    float a = sqr(Rs) + 1.0f, b2 = 2.0f*Rs, c = sqr(Rp) + 1.0f, d2 = 2.0f*Rp, ts = sqr(Ts), tp = sqr(Tp);
    f3 tS = mk3(ts/(a - b2*cosPhi.x), ts/(a - b2*cosPhi.y), ts/(a - b2*cosPhi.z));
    f3 tP = mk3(tp/(c - d2*cosPhi.x), tp/(c - d2*cosPhi.y), tp/(c - d2*cosPhi.z));
    return mk3(1.0f - (tS.x + tP.x)*0.5f, 1.0f - (tS.y + tP.y)*0.5f, 1.0f - (tS.z + tP.z)*0.5f);
}
PT_DEV float dielectricReflectance(float eta, float cosThetaI) { float t; return dielectricReflectance(eta, cosThetaI, t); }

PT_DEV float conductorReflectance1(float eta, float k, float cosThetaI)
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
PT_DEV f3 conductorReflectance(const float *eta, const float *k, float cosThetaI)
{
    return mk3(conductorReflectance1(eta[0], k[0], cosThetaI), conductorReflectance1(eta[1], k[1], cosThetaI),
               conductorReflectance1(eta[2], k[2], cosThetaI));
}

// The distribution of a microfacet BSDF as a shading variant sees it.  Phong's D and sample() go through pow(double, double) (Microfacet.hpp:
// 49-51, 100-102: 3 200 double-precision instructions and 39 VGPRs in the conductor-family variant); only the variants with FEAT_PHONG
// carry them -- the shim puts materials with a Phong distribution into class 3 (tungsten_hip.hip: bsdfTypeMask), so the family variants
// never see one and the compiler drops the branch.
template<uint32_t M> PT_DEV int mfDist(int d) { return (M & FEAT_PHONG) ? d : (d == TGHIP_DIST_BECKMANN ? TGHIP_DIST_BECKMANN : TGHIP_DIST_GGX); }
PT_DEV float mfRoughnessToAlpha(int dist, float roughness)
{
    roughness = fmaxf(roughness, 1e-3f);
    if (dist == TGHIP_DIST_PHONG)
        return 2.0f/(roughness*roughness) - 2.0f;
    return roughness;
}
PT_DEV float mfD(int dist, float alpha, f3 m)
{
    if (m.z <= 0.0f)
        return 0.0f;
    if (dist == TGHIP_DIST_PHONG)
        return (alpha + 2.0f)*PT_INV_TWO_PI*(float)pow((double)m.z, (double)alpha);
    float alphaSq = alpha*alpha;
    float cosThetaSq = m.z*m.z;
    float tanThetaSq = fmaxf(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
    float cosThetaQu = cosThetaSq*cosThetaSq;
    if (dist == TGHIP_DIST_BECKMANN)
        return PT_INV_PI*expfH(-tanThetaSq/alphaSq)/(alphaSq*cosThetaQu);
    return alphaSq*PT_INV_PI/(cosThetaQu*sqr(alphaSq + tanThetaSq));
}
PT_DEV float mfG1(int dist, float alpha, f3 v, f3 m)
{
    if (dot(v, m)*v.z <= 0.0f)
        return 0.0f;
    float cosThetaSq = v.z*v.z;
    if (dist == TGHIP_DIST_GGX) {
        float alphaSq = alpha*alpha;
        float tanThetaSq = fmaxf(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
        return 2.0f/(1.0f + sqrtf(1.0f + alphaSq*tanThetaSq));
    }
    float tanTheta = fabsf(sqrtf(fmaxf(1.0f - cosThetaSq, 0.0f))/v.z);
    float a = dist == TGHIP_DIST_BECKMANN ? 1.0f/(alpha*tanTheta) : sqrtf(0.5f*alpha + 1.0f)/tanTheta;
    if (a < 1.6f)
        return (3.535f*a + 2.181f*a*a)/(1.0f + 2.276f*a + 2.577f*a*a);
    return 1.0f;
}
PT_DEV float mfG(int dist, float alpha, f3 i, f3 o, f3 m) { return mfG1(dist, alpha, i, m)*mfG1(dist, alpha, o, m); }
PT_DEV float mfPdf(int dist, float alpha, f3 m) { return mfD(dist, alpha, m)*m.z; }
PT_DEV f3 mfSample(int dist, float alpha, float xi0, float xi1)
{
    float phi = xi1*PT_TWO_PI;
    float cosTheta;
    if (dist == TGHIP_DIST_BECKMANN) {
        float tanThetaSq = -alpha*alpha*logfH(1.0f - xi0);
        cosTheta = 1.0f/sqrtf(1.0f + tanThetaSq);
    } else if (dist == TGHIP_DIST_PHONG) {
        cosTheta = (float)pow((double)xi0, 1.0/((double)alpha + 2.0));
    } else {
        float tanThetaSq = alpha*alpha*xi0/(1.0f - xi0);
        cosTheta = 1.0f/sqrtf(1.0f + tanThetaSq);
    }
    float r = sqrtf(fmaxf(1.0f - cosTheta*cosTheta, 0.0f));
    float sinPhi, cosPhi;
    sincosfH(phi, sinPhi, cosPhi);
    return mk3(cosPhi*r, sinPhi*r, cosTheta);
}

// ---------------------------------------------------------------------------------------------
// BSDF interpreter.  Event = the mutable part of SurfaceScatterEvent (SurfaceScatterEvent.hpp:14-44)
// ---------------------------------------------------------------------------------------------
struct Event {
    f3 wi, wo, weight;
    float pdf;
    uint32_t requested, sampled;
    float u, v;
    Rng *rng;
};

PT_DEV bool checkReflectionConstraint(f3 wi, f3 wo)      /* Bsdf.hpp:45-48 */
{
    return fabsf(wi.z*wo.z - wi.x*wo.x - wi.y*wo.y - 1.0f) < 1e-3f;
}
PT_DEV bool checkRefractionConstraint(f3 wi, f3 wo, float eta, float cosThetaT)   /* Bsdf.hpp:50-54 */
{
    float dotP = -wi.x*wo.x*eta - wi.y*wo.y*eta - copysignf(cosThetaT, wi.z)*wo.z;
    return fabsf(dotP - 1.0f) < 1e-3f;
}
PT_DEV float sgnE(float v) { return v < 0.0f ? -1.0f : 1.0f; }
PT_DEV bool isExactReverse(f3 wi, f3 wo) { return -wi.x == wo.x && -wi.y == wo.y && -wi.z == wo.z; }

template<uint32_t M> PT_DEV f3 bsdfAlbedo(const DeviceScene &s, const TgHipBsdf &b, const Event &e) { return textureEval<M>(s, b.albedo, e.u, e.v); }
template<uint32_t M> PT_DEV float bsdfRoughness(const DeviceScene &s, const TgHipBsdf &b, const Event &e) { return textureEval<M>(s, b.roughness, e.u, e.v).x; }

// RoughDielectricBsdf::sampleBase / evalBase / pdfBase (RoughDielectricBsdf.cpp:55-131,133-166,200-236)
template<uint32_t M> PT_DEV bool rdSampleBase(Event &e, bool sampleR, bool sampleT, float roughness, float ior, int dist)
{
    float wiDotN = e.wi.z;
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    float sampleRoughness = (1.2f - 0.2f*sqrtf(fabsf(wiDotN)))*roughness;
    float alpha = mfRoughnessToAlpha(dist, roughness);
    float sampleAlpha = mfRoughnessToAlpha(dist, sampleRoughness);
    float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
    f3 m = mfSample(dist, sampleAlpha, xi0, xi1);
    float pm = mfPdf(dist, sampleAlpha, m);
    if (pm < 1e-10f)
        return false;
    float wiDotM = dot(e.wi, m);
    float cosThetaT = 0.0f;
    float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
    float etaM = wiDotM < 0.0f ? ior : 1.0f/ior;
    bool reflect;
    if (sampleR && sampleT) {
        reflect = rngNextBoolean(*e.rng, F);
    } else if (sampleT) {
        if (F == 1.0f)
            return false;
        reflect = false;
    } else if (sampleR) {
        reflect = true;
    } else {
        return false;
    }
    if (reflect)
        e.wo = m*(2.0f*wiDotM) - e.wi;
    else
        e.wo = m*(etaM*wiDotM - sgnE(wiDotM)*cosThetaT) - e.wi*etaM;
    float woDotN = e.wo.z;
    bool reflected = wiDotN*woDotN > 0.0f;
    if (reflected != reflect)
        return false;
    float woDotM = dot(e.wo, m);
    float G = mfG(dist, alpha, e.wi, e.wo, m);
    float D = mfD(dist, alpha, m);
    e.weight = splat3(fabsf(wiDotM)*G*D/(fabsf(wiDotN)*pm));
    if (reflect) {
        e.pdf = pm*0.25f/fabsf(wiDotM);
        e.sampled = TGHIP_LOBE_GLOSSY_R;
    } else {
        e.pdf = pm*fabsf(woDotM)/sqr(eta*wiDotM + woDotM);
        e.sampled = TGHIP_LOBE_GLOSSY_T;
    }
    if (sampleR && sampleT) {
        e.pdf *= reflect ? F : 1.0f - F;
    } else {
        e.weight = e.weight*(reflect ? F : 1.0f - F);
    }
    return true;
}
PT_DEV f3 rdHalfVector(const Event &e, bool reflect, float eta)
{
    if (reflect)
        return normalized(e.wi + e.wo)*sgnE(e.wi.z);
    return -normalized(e.wi*eta + e.wo);
}
PT_DEV f3 rdEvalBase(const Event &e, bool sampleR, bool sampleT, float roughness, float ior, int dist)
{
    float wiDotN = e.wi.z, woDotN = e.wo.z;
    bool reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT))
        return splat3(0.0f);
    float alpha = mfRoughnessToAlpha(dist, roughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    f3 m = rdHalfVector(e, reflect, eta);
    float wiDotM = dot(e.wi, m), woDotM = dot(e.wo, m);
    float F = dielectricReflectance(1.0f/ior, wiDotM);
    float G = mfG(dist, alpha, e.wi, e.wo, m);
    float D = mfD(dist, alpha, m);
    if (reflect)
        return splat3((F*G*D*0.25f)/fabsf(wiDotN));
    return splat3(fabsf(wiDotM*woDotM)*(1.0f - F)*G*D/(sqr(eta*wiDotM + woDotM)*fabsf(wiDotN)));
}
PT_DEV float rdPdfBase(const Event &e, bool sampleR, bool sampleT, float roughness, float ior, int dist)
{
    float wiDotN = e.wi.z, woDotN = e.wo.z;
    bool reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT))
        return 0.0f;
    float sampleRoughness = (1.2f - 0.2f*sqrtf(fabsf(wiDotN)))*roughness;
    float sampleAlpha = mfRoughnessToAlpha(dist, sampleRoughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    f3 m = rdHalfVector(e, reflect, eta);
    float wiDotM = dot(e.wi, m), woDotM = dot(e.wo, m);
    float F = dielectricReflectance(1.0f/ior, wiDotM);
    float pm = mfPdf(dist, sampleAlpha, m);
    float pdf = reflect ? pm*0.25f/fabsf(wiDotM) : pm*fabsf(woDotM)/sqr(eta*wiDotM + woDotM);
    if (sampleR && sampleT)
        pdf *= reflect ? F : 1.0f - F;
    return pdf;
}

PT_DEV f3 plasticSubstrate(const TgHipBsdf &b, f3 diffuseAlbedo)
{
    return diffuseAlbedo/(splat3(1.0f) - diffuseAlbedo*b.diffuse_fresnel);
}
PT_DEV f3 absorb(const TgHipBsdf &b, f3 f, float cosA, float cosB)
{
    f3 ssa = ld3(b.scaled_sigma_a);
    if (max3(ssa) > 0.0f)
        return f*exp3(ssa*(-1.0f/cosA - 1.0f/cosB));
    return f;
}

template<int D, uint32_t M> struct BsdfOps;

template<int D, uint32_t M>
struct BsdfOps {
    typedef BsdfOps<D + 1, M> Next;

    static __device__ f3 eval(const DeviceScene &s, int bi, const Event &e)
    {
        const TgHipBsdf &b = s.bsdfs[bi];
        switch (b.type) {
        case TGHIP_BSDF_LAMBERT: case TGHIP_BSDF_ERROR:        /* LambertBsdf.cpp:40-47 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_LAMBERT) | BSDF_BIT(TGHIP_BSDF_ERROR)))) return splat3(0.0f);
            if (!(e.requested & TGHIP_LOBE_DIFFUSE_R)) return splat3(0.0f);
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat3(0.0f);
            return bsdfAlbedo<M>(s, b, e)*PT_INV_PI*e.wo.z;
        case TGHIP_BSDF_FORWARD:                               /* ForwardBsdf.cpp:25-28 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_FORWARD)))) return splat3(0.0f);
            return (e.requested == TGHIP_LOBE_FORWARD && isExactReverse(e.wi, e.wo)) ? splat3(1.0f) : splat3(0.0f);
        case TGHIP_BSDF_MIRROR:                                /* MirrorBsdf.cpp:39-46 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_MIRROR)))) return splat3(0.0f);
            if ((e.requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e.wi, e.wo))
                return bsdfAlbedo<M>(s, b, e);
            return splat3(0.0f);
        case TGHIP_BSDF_CONDUCTOR:                             /* ConductorBsdf.cpp:68-75 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_CONDUCTOR)))) return splat3(0.0f);
            if ((e.requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e.wi, e.wo))
                return bsdfAlbedo<M>(s, b, e)*conductorReflectance(b.eta, b.k, e.wi.z);
            return splat3(0.0f);
        case TGHIP_BSDF_ROUGH_CONDUCTOR: {                     /* RoughConductorBsdf.cpp:93-109 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_CONDUCTOR)))) return splat3(0.0f);
            if (!(e.requested & TGHIP_LOBE_GLOSSY_R)) return splat3(0.0f);
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat3(0.0f);
            float alpha = mfRoughnessToAlpha(mfDist<M>(b.distribution), bsdfRoughness<M>(s, b, e));
            f3 hr = normalized(e.wi + e.wo);
            float cosThetaM = dot(e.wi, hr);
            f3 F = conductorReflectance(b.eta, b.k, cosThetaM);
            float G = mfG(mfDist<M>(b.distribution), alpha, e.wi, e.wo, hr);
            float Dm = mfD(mfDist<M>(b.distribution), alpha, hr);
            float fr = (G*Dm*0.25f)/e.wi.z;
            return bsdfAlbedo<M>(s, b, e)*(F*fr);
        }
        case TGHIP_BSDF_SMOOTH_COAT: {                         /* SmoothCoatBsdf.cpp:146-177 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_SMOOTH_COAT)))) return splat3(0.0f);
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat3(0.0f);
            bool evalR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool evalT = (e.requested & s.bsdfs[b.sub0].lobes) != 0;
            float eta = 1.0f/b.ior;
            float cosThetaTi, cosThetaTo;
            float Fi = dielectricReflectance(eta, e.wi.z, cosThetaTi);
            float Fo = dielectricReflectance(eta, e.wo.z, cosThetaTo);
            if (evalR && checkReflectionConstraint(e.wi, e.wo))
                return splat3(Fi);
            if (evalT) {
                Event q = e;
                q.wi = mk3(e.wi.x*eta, e.wi.y*eta, copysignf(cosThetaTi, e.wi.z));
                q.wo = mk3(e.wo.x*eta, e.wo.y*eta, copysignf(cosThetaTo, e.wo.z));
                float laplacian = eta*eta*e.wo.z/cosThetaTo;
                f3 substrateF = absorb(b, Next::eval(s, b.sub0, q), cosThetaTo, cosThetaTi);
                return substrateF*(laplacian*(1.0f - Fi)*(1.0f - Fo));
            }
            return splat3(0.0f);
        }
        case TGHIP_BSDF_DIELECTRIC: {                          /* DielectricBsdf.cpp:88-108 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_DIELECTRIC)))) return splat3(0.0f);
            bool evalR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool evalT = (e.requested & TGHIP_LOBE_SPECULAR_T) && b.enable_refraction;
            float eta = e.wi.z < 0.0f ? b.ior : 1.0f/b.ior;
            float cosThetaT = 0.0f;
            float F = dielectricReflectance(eta, fabsf(e.wi.z), cosThetaT);
            if (e.wi.z*e.wo.z >= 0.0f) {
                if (evalR && checkReflectionConstraint(e.wi, e.wo))
                    return bsdfAlbedo<M>(s, b, e)*F;
                return splat3(0.0f);
            }
            if (evalT && checkRefractionConstraint(e.wi, e.wo, eta, cosThetaT))
                return bsdfAlbedo<M>(s, b, e)*(1.0f - F);
            return splat3(0.0f);
        }
        case TGHIP_BSDF_ROUGH_DIELECTRIC: {                    /* RoughDielectricBsdf.cpp:247-254 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_DIELECTRIC)))) return splat3(0.0f);
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_GLOSSY_T) && b.enable_refraction;
            return rdEvalBase(e, sampleR, sampleT, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution))*bsdfAlbedo<M>(s, b, e);
        }
        case TGHIP_BSDF_PLASTIC: case TGHIP_BSDF_ROUGH_PLASTIC: {   /* PlasticBsdf.cpp:125-151, RoughPlasticBsdf.cpp:114-141 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_PLASTIC) | BSDF_BIT(TGHIP_BSDF_ROUGH_PLASTIC)))) return splat3(0.0f);
            bool rough = b.type == TGHIP_BSDF_ROUGH_PLASTIC;
            bool evalR = (e.requested & (rough ? TGHIP_LOBE_GLOSSY_R : TGHIP_LOBE_SPECULAR_R)) != 0;
            bool evalT = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (rough && !evalR && !evalT) return splat3(0.0f);
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat3(0.0f);
            float eta = 1.0f/b.ior;
            float Fi = dielectricReflectance(eta, e.wi.z);
            float Fo = dielectricReflectance(eta, e.wo.z);
            if (!rough) {
                if (evalR && checkReflectionConstraint(e.wi, e.wo))
                    return splat3(Fi);
                if (!evalT)
                    return splat3(0.0f);
            }
            f3 glossyR = splat3(0.0f), diffuseR = splat3(0.0f);
            if (rough && evalR)
                glossyR = rdEvalBase(e, true, false, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution));
            if (evalT) {
                f3 diffuseAlbedo = bsdfAlbedo<M>(s, b, e);
                diffuseR = plasticSubstrate(b, diffuseAlbedo)*((1.0f - Fi)*(1.0f - Fo)*eta*eta*e.wo.z*PT_INV_PI);
                diffuseR = absorb(b, diffuseR, e.wo.z, e.wi.z);
            }
            return rough ? glossyR + diffuseR : diffuseR;
        }
        case TGHIP_BSDF_MIXED: {                               /* MixedBsdf.cpp:101-105 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_MIXED)))) return splat3(0.0f);
            float ratio = textureEval<M>(s, b.tex1, e.u, e.v).x;
            f3 f0 = Next::eval(s, b.sub0, e), f1 = Next::eval(s, b.sub1, e);
            return bsdfAlbedo<M>(s, b, e)*(f0*ratio + f1*(1.0f - ratio));
        }
        case TGHIP_BSDF_DIFFUSE_TRANSMISSION: {                /* DiffuseTransmissionBsdf.cpp:50-57 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_DIFFUSE_TRANSMISSION))) return splat3(0.0f);
            if (!(e.requested & TGHIP_LOBE_DIFFUSE_T)) return splat3(0.0f);
            float factor = e.wi.z*e.wo.z < 0.0f ? b.eta[0] : 1.0f - b.eta[0];
            return bsdfAlbedo<M>(s, b, e)*factor*PT_INV_PI*fabsf(e.wo.z);
        }
        case TGHIP_BSDF_PHONG: {                               /* PhongBsdf.cpp:79-99 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_PHONG))) return splat3(0.0f);
            bool evalGlossy = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0, evalDiffuse = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (!evalGlossy && !evalDiffuse) return splat3(0.0f);
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat3(0.0f);
            float result = 0.0f;
            if (evalDiffuse)
                result += b.eta[1]*PT_INV_PI;
            if (evalGlossy) {
                float cosTheta = dot(mk3(-e.wi.x, -e.wi.y, e.wi.z), e.wo);
                if (cosTheta > 0.0f)
                    result += powfH(cosTheta, b.eta[0])*b.k[2]*(1.0f - b.eta[1]);
            }
            return bsdfAlbedo<M>(s, b, e)*e.wo.z*result;
        }
        case TGHIP_BSDF_THINSHEET: {                           /* ThinSheetBsdf.cpp:83-104 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_THINSHEET))) return splat3(0.0f);
            if (e.requested != TGHIP_LOBE_FORWARD || !isExactReverse(e.wi, e.wo))
                return splat3(0.0f);
            float thickness = textureEval<M>(s, b.tex1, e.u, e.v).x;
            float cosThetaT;
            f3 transmittance;
            if (b.enable_refraction)                           /* _enableInterference */
                transmittance = splat3(1.0f) - thinFilmReflectanceInterference(1.0f/b.ior, fabsf(e.wi.z), thickness*500.0f, cosThetaT);
            else
                transmittance = splat3(1.0f - thinFilmReflectance(1.0f/b.ior, fabsf(e.wi.z), cosThetaT));
            f3 sa = ld3(b.sigma_a);
            if (!(sa.x == 0.0f && sa.y == 0.0f && sa.z == 0.0f) && cosThetaT > 0.0f)
                transmittance = transmittance*exp3((-sa)*(thickness*2.0f/cosThetaT));
            return transmittance;
        }
        case TGHIP_BSDF_OREN_NAYAR: {                          /* OrenNayarBsdf.cpp:61-100 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_OREN_NAYAR))) return splat3(0.0f);
            if (!(e.requested & TGHIP_LOBE_DIFFUSE_R)) return splat3(0.0f);
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat3(0.0f);
            const f3 wi = e.wi, wo = e.wo;
            float thetaR = acosfExact(wo.z);
            float thetaI = acosfExact(wi.z);
            float alpha = thetaR > thetaI ? thetaR : thetaI;
            float beta = thetaR < thetaI ? thetaR : thetaI;
            float sinAlpha = sinfH(alpha);
            float denom = (wi.x*wi.x + wi.y*wi.y)*(wo.x*wo.x + wo.y*wo.y);
            float cosDeltaPhi;
            if (denom == 0.0f)
                cosDeltaPhi = 1.0f;
            else
                cosDeltaPhi = (wi.x*wo.x + wi.y*wo.y)/sqrtf(denom);
            const float RoughnessToSigma = 1.0f/sqrtf(2.0f);
            float sigma = RoughnessToSigma*bsdfRoughness<M>(s, b, e);
            float sigmaSq = sigma*sigma;
            float C1 = 1.0f - 0.5f*sigmaSq/(sigmaSq + 0.33f);
            float C2 = 0.45f*sigmaSq/(sigmaSq + 0.09f);
            if (cosDeltaPhi >= 0.0f) {
                C2 *= sinAlpha;
            } else {
                float q = (2.0f*PT_INV_PI)*beta;
                C2 *= sinAlpha - q*q*q;
            }
            float C3 = 0.125f*(sigmaSq/(sigmaSq + 0.09f))*sqr((4.0f*PT_INV_PI*PT_INV_PI)*alpha*beta);
            float fr1 = (C1 + cosDeltaPhi*C2*tanfH(beta) + (1.0f - fabsf(cosDeltaPhi))*C3*tanfH(0.5f*(alpha + beta)));
            float fr2 = 0.17f*sigmaSq/(sigmaSq + 0.13f)*(1.0f - cosDeltaPhi*sqr((2.0f*PT_INV_PI)*beta));
            f3 diffuseAlbedo = bsdfAlbedo<M>(s, b, e);
            return (diffuseAlbedo*fr1 + diffuseAlbedo*diffuseAlbedo*fr2)*wo.z*PT_INV_PI;
        }
        case TGHIP_BSDF_ROUGH_COAT: {                          /* RoughCoatBsdf.cpp:161-199 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_ROUGH_COAT))) return splat3(0.0f);
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & s.bsdfs[b.sub0].lobes) != 0;
            if (!sampleT && !sampleR) return splat3(0.0f);
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat3(0.0f);
            f3 glossyR = splat3(0.0f);
            if (sampleR)
                glossyR = rdEvalBase(e, true, false, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution));
            f3 substrateR = splat3(0.0f);
            if (sampleT) {
                float eta = 1.0f/b.ior;
                float cosThetaTi, cosThetaTo;
                float Fi = dielectricReflectance(eta, e.wi.z, cosThetaTi);
                float Fo = dielectricReflectance(eta, e.wo.z, cosThetaTo);
                if (Fi == 1.0f || Fo == 1.0f)
                    return glossyR;
                Event q = e;
                q.wi = mk3(e.wi.x*eta, e.wi.y*eta, copysignf(cosThetaTi, e.wi.z));
                q.wo = mk3(e.wo.x*eta, e.wo.y*eta, copysignf(cosThetaTo, e.wo.z));
                float compressionProjection = eta*eta*e.wo.z/cosThetaTo;
                f3 substrateF = absorb(b, Next::eval(s, b.sub0, q), cosThetaTo, cosThetaTi);
                substrateR = substrateF*(compressionProjection*(1.0f - Fi)*(1.0f - Fo));
            }
            return glossyR + substrateR;
        }
        case TGHIP_BSDF_TRANSPARENCY:                          /* TransparencyBsdf.cpp:48-54 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_TRANSPARENCY)))) return splat3(0.0f);
            if (e.requested == TGHIP_LOBE_FORWARD)
                return isExactReverse(e.wi, e.wo) ? splat3(1.0f - textureEval<M>(s, b.tex1, e.u, e.v).x) : splat3(0.0f);
            return Next::eval(s, b.sub0, e);
        default:
            return splat3(0.0f);
        }
    }

    static __device__ bool mixedRatio(const DeviceScene &s, const TgHipBsdf &b, const Event &e, float &ratio)   /* MixedBsdf.cpp:17-31 */
    {
        bool sample0 = (e.requested & s.bsdfs[b.sub0].lobes) != 0;
        bool sample1 = (e.requested & s.bsdfs[b.sub1].lobes) != 0;
        if (sample0 && sample1) ratio = textureEval<M>(s, b.tex1, e.u, e.v).x;
        else if (sample0) ratio = 1.0f;
        else if (sample1) ratio = 0.0f;
        else return false;
        return true;
    }

    static __device__ bool sample(const DeviceScene &s, int bi, Event &e)
    {
        const TgHipBsdf &b = s.bsdfs[bi];
        switch (b.type) {
        case TGHIP_BSDF_LAMBERT: case TGHIP_BSDF_ERROR: {      /* LambertBsdf.cpp:27-38 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_LAMBERT) | BSDF_BIT(TGHIP_BSDF_ERROR)))) return false;
            if (!(e.requested & TGHIP_LOBE_DIFFUSE_R)) return false;
            if (e.wi.z <= 0.0f) return false;
            float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
            e.wo = cosineHemisphere(xi0, xi1);
            e.pdf = cosineHemispherePdf(e.wo);
            e.weight = bsdfAlbedo<M>(s, b, e);
            e.sampled = TGHIP_LOBE_DIFFUSE_R;
            return true;
        }
        case TGHIP_BSDF_MIRROR:                                /* MirrorBsdf.cpp:28-37 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_MIRROR)))) return false;
            if (!(e.requested & TGHIP_LOBE_SPECULAR_R)) return false;
            e.wo = mk3(-e.wi.x, -e.wi.y, e.wi.z);
            e.pdf = 1.0f;
            e.sampled = TGHIP_LOBE_SPECULAR_R;
            e.weight = bsdfAlbedo<M>(s, b, e);
            return true;
        case TGHIP_BSDF_CONDUCTOR:                             /* ConductorBsdf.cpp:56-66 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_CONDUCTOR)))) return false;
            if (!(e.requested & TGHIP_LOBE_SPECULAR_R)) return false;
            e.wo = mk3(-e.wi.x, -e.wi.y, e.wi.z);
            e.pdf = 1.0f;
            e.weight = bsdfAlbedo<M>(s, b, e)*conductorReflectance(b.eta, b.k, e.wi.z);
            e.sampled = TGHIP_LOBE_SPECULAR_R;
            return true;
        case TGHIP_BSDF_ROUGH_CONDUCTOR: {                     /* RoughConductorBsdf.cpp:60-91 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_CONDUCTOR)))) return false;
            if (!(e.requested & TGHIP_LOBE_GLOSSY_R)) return false;
            if (e.wi.z <= 0.0f) return false;
            float alpha = mfRoughnessToAlpha(mfDist<M>(b.distribution), bsdfRoughness<M>(s, b, e));
            float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
            f3 m = mfSample(mfDist<M>(b.distribution), alpha, xi0, xi1);
            float wiDotM = dot(e.wi, m);
            e.wo = m*(2.0f*wiDotM) - e.wi;
            if (wiDotM <= 0.0f || e.wo.z <= 0.0f)
                return false;
            float G = mfG(mfDist<M>(b.distribution), alpha, e.wi, e.wo, m);
            float Dm = mfD(mfDist<M>(b.distribution), alpha, m);
            float mPdf = mfPdf(mfDist<M>(b.distribution), alpha, m);
            float weight = wiDotM*G*Dm/(e.wi.z*mPdf);
            f3 F = conductorReflectance(b.eta, b.k, wiDotM);
            e.pdf = mPdf*0.25f/wiDotM;
            e.weight = bsdfAlbedo<M>(s, b, e)*(F*weight);
            e.sampled = TGHIP_LOBE_GLOSSY_R;
            return true;
        }
        case TGHIP_BSDF_SMOOTH_COAT: {                         /* SmoothCoatBsdf.cpp:41-100 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_SMOOTH_COAT)))) return false;
            if (e.wi.z <= 0.0f) return false;
            bool sampleR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool sampleT = (e.requested & s.bsdfs[b.sub0].lobes) != 0;
            if (!sampleR && !sampleT) return false;
            f3 wi = e.wi;
            float eta = 1.0f/b.ior;
            float cosThetaTi;
            float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
            float substrateWeight = b.avg_transmittance*(1.0f - Fi);
            float specularProbability = (sampleR && sampleT) ? Fi/(Fi + substrateWeight) : (sampleR ? 1.0f : 0.0f);
            if (sampleR && rngNextBoolean(*e.rng, specularProbability)) {
                e.wo = mk3(-wi.x, -wi.y, wi.z);
                e.pdf = specularProbability;
                e.weight = splat3(Fi/specularProbability);
                e.sampled = TGHIP_LOBE_SPECULAR_R;
                return true;
            }
            e.wi = mk3(wi.x*eta, wi.y*eta, cosThetaTi);
            bool success = Next::sample(s, b.sub0, e);
            e.wi = wi;
            if (!success) return false;
            float cosThetaTo;
            float Fo = dielectricReflectance(b.ior, e.wo.z, cosThetaTo);
            if (Fo == 1.0f) return false;
            float cosThetaSubstrate = e.wo.z;
            e.wo = mk3(e.wo.x*b.ior, e.wo.y*b.ior, cosThetaTo);
            e.weight = e.weight*((1.0f - Fi)*(1.0f - Fo));
            e.weight = absorb(b, e.weight, cosThetaSubstrate, cosThetaTi);
            e.weight = e.weight/(1.0f - specularProbability);
            e.pdf *= 1.0f - specularProbability;
            e.pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
            return true;
        }
        case TGHIP_BSDF_DIELECTRIC: {                          /* DielectricBsdf.cpp:49-86 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_DIELECTRIC)))) return false;
            bool sampleR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_SPECULAR_T) && b.enable_refraction;
            float eta = e.wi.z < 0.0f ? b.ior : 1.0f/b.ior;
            float cosThetaT = 0.0f;
            float F = dielectricReflectance(eta, fabsf(e.wi.z), cosThetaT);
            float reflectionProbability;
            if (sampleR && sampleT) reflectionProbability = F;
            else if (sampleR) reflectionProbability = 1.0f;
            else if (sampleT) reflectionProbability = 0.0f;
            else return false;
            if (rngNextBoolean(*e.rng, reflectionProbability)) {
                e.wo = mk3(-e.wi.x, -e.wi.y, e.wi.z);
                e.pdf = reflectionProbability;
                e.sampled = TGHIP_LOBE_SPECULAR_R;
                e.weight = sampleT ? splat3(1.0f) : splat3(F);
            } else {
                if (F == 1.0f) return false;
                e.wo = mk3(-e.wi.x*eta, -e.wi.y*eta, -copysignf(cosThetaT, e.wi.z));
                e.pdf = 1.0f - reflectionProbability;
                e.sampled = TGHIP_LOBE_SPECULAR_T;
                e.weight = sampleR ? splat3(1.0f) : splat3(1.0f - F);
            }
            e.weight = e.weight*bsdfAlbedo<M>(s, b, e);
            return true;
        }
        case TGHIP_BSDF_ROUGH_DIELECTRIC: {                    /* RoughDielectricBsdf.cpp:238-245 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_DIELECTRIC)))) return false;
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_GLOSSY_T) && b.enable_refraction;
            bool result = rdSampleBase<M>(e, sampleR, sampleT, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution));
            e.weight = e.weight*bsdfAlbedo<M>(s, b, e);
            return result;
        }
        case TGHIP_BSDF_PLASTIC: {                             /* PlasticBsdf.cpp:45-87 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_PLASTIC)))) return false;
            if (e.wi.z <= 0.0f) return false;
            bool sampleR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (!sampleR && !sampleT) return false;
            f3 wi = e.wi;
            float eta = 1.0f/b.ior;
            float Fi = dielectricReflectance(eta, wi.z);
            float substrateWeight = b.avg_transmittance*(1.0f - Fi);
            float specularProbability = (sampleR && sampleT) ? Fi/(Fi + substrateWeight) : (sampleR ? 1.0f : 0.0f);
            if (sampleR && rngNextBoolean(*e.rng, specularProbability)) {
                e.wo = mk3(-wi.x, -wi.y, wi.z);
                e.pdf = specularProbability;
                e.weight = splat3(Fi/specularProbability);
                e.sampled = TGHIP_LOBE_SPECULAR_R;
            } else {
                float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
                f3 wo = cosineHemisphere(xi0, xi1);
                float Fo = dielectricReflectance(eta, wo.z);
                e.wo = wo;
                e.weight = plasticSubstrate(b, bsdfAlbedo<M>(s, b, e))*((1.0f - Fi)*(1.0f - Fo)*eta*eta);
                e.weight = absorb(b, e.weight, e.wo.z, e.wi.z);
                e.pdf = cosineHemispherePdf(e.wo)*(1.0f - specularProbability);
                e.weight = e.weight/(1.0f - specularProbability);
                e.sampled = TGHIP_LOBE_DIFFUSE_R;
            }
            return true;
        }
        case TGHIP_BSDF_ROUGH_PLASTIC: {                       /* RoughPlasticBsdf.cpp:54-112 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_PLASTIC)))) return false;
            if (e.wi.z <= 0.0f) return false;
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (!sampleR && !sampleT) return false;
            float eta = 1.0f/b.ior;
            float Fi = dielectricReflectance(eta, e.wi.z);
            float substrateW = avg3(ld3(s.textures[b.albedo].avg));
            float substrateWeight = substrateW*b.avg_transmittance*(1.0f - Fi);
            float specularProbability = Fi/(Fi + substrateWeight);
            if (sampleR && (rngNextBoolean(*e.rng, specularProbability) || !sampleT)) {
                if (!rdSampleBase<M>(e, true, false, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution)))
                    return false;
                if (sampleT) {
                    float Fo = dielectricReflectance(eta, e.wo.z);
                    f3 brdfSubstrate = plasticSubstrate(b, bsdfAlbedo<M>(s, b, e))*((1.0f - Fi)*(1.0f - Fo)*eta*eta)*PT_INV_PI*e.wo.z;
                    f3 brdfSpecular = e.weight*e.pdf;
                    float pdfSubstrate = cosineHemispherePdf(e.wo)*(1.0f - specularProbability);
                    float pdfSpecular = e.pdf*specularProbability;
                    e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
                    e.pdf = pdfSpecular + pdfSubstrate;
                }
                return true;
            }
            float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
            f3 wo = cosineHemisphere(xi0, xi1);
            float Fo = dielectricReflectance(eta, wo.z);
            e.wo = wo;
            e.weight = plasticSubstrate(b, bsdfAlbedo<M>(s, b, e))*((1.0f - Fi)*(1.0f - Fo)*eta*eta);
            e.weight = absorb(b, e.weight, e.wo.z, e.wi.z);
            e.pdf = cosineHemispherePdf(e.wo);
            if (sampleR) {
                f3 brdfSubstrate = e.weight*e.pdf;
                float pdfSubstrate = e.pdf*(1.0f - specularProbability);
                float r = bsdfRoughness<M>(s, b, e);
                f3 brdfSpecular = rdEvalBase(e, true, false, r, b.ior, mfDist<M>(b.distribution));
                float pdfSpecular = rdPdfBase(e, true, false, r, b.ior, mfDist<M>(b.distribution))*specularProbability;
                e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
                e.pdf = pdfSpecular + pdfSubstrate;
            }
            e.sampled = TGHIP_LOBE_DIFFUSE_R;
            return true;
        }
        case TGHIP_BSDF_MIXED: {                               /* MixedBsdf.cpp:70-99 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_MIXED)))) return false;
            float ratio;
            if (!mixedRatio(s, b, e, ratio)) return false;
            if (rngNextBoolean(*e.rng, ratio)) {
                if (!Next::sample(s, b.sub0, e)) return false;
                float pdf0 = e.pdf*ratio;
                float pdf1 = Next::pdf(s, b.sub1, e)*(1.0f - ratio);
                f3 f = e.weight*e.pdf*ratio + Next::eval(s, b.sub1, e)*(1.0f - ratio);
                e.pdf = pdf0 + pdf1;
                e.weight = f/e.pdf;
            } else {
                if (!Next::sample(s, b.sub1, e)) return false;
                float pdf0 = Next::pdf(s, b.sub0, e)*ratio;
                float pdf1 = e.pdf*(1.0f - ratio);
                f3 f = Next::eval(s, b.sub0, e)*ratio + e.weight*e.pdf*(1.0f - ratio);
                e.pdf = pdf0 + pdf1;
                e.weight = f/e.pdf;
            }
            e.weight = e.weight*bsdfAlbedo<M>(s, b, e);
            return true;
        }
        case TGHIP_BSDF_DIFFUSE_TRANSMISSION: {                /* DiffuseTransmissionBsdf.cpp:29-48 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_DIFFUSE_TRANSMISSION))) return false;
            bool sampleR = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0, sampleT = (e.requested & TGHIP_LOBE_DIFFUSE_T) != 0;
            if (!sampleR && !sampleT) return false;
            const float T = b.eta[0];
            float transmittanceProbability = sampleR && sampleT ? T : (sampleR ? 0.0f : 1.0f);
            bool transmit = rngNextBoolean(*e.rng, transmittanceProbability);
            float weight = sampleR && sampleT ? 1.0f : (transmit ? T : 1.0f - T);
            float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
            e.wo = cosineHemisphere(xi0, xi1);
            e.wo.z = copysignf(e.wo.z, e.wi.z);
            if (transmit)
                e.wo.z = -e.wo.z;
            e.pdf = cosineHemispherePdf(e.wo);
            e.weight = bsdfAlbedo<M>(s, b, e)*weight;
            e.sampled = TGHIP_LOBE_DIFFUSE_T;
            return true;
        }
        case TGHIP_BSDF_PHONG: {                               /* PhongBsdf.cpp:39-77 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_PHONG))) return false;
            bool evalGlossy = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0, evalDiffuse = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (!evalGlossy && !evalDiffuse) return false;
            if (e.wi.z <= 0.0f) return false;
            bool sampleGlossy;
            if (evalGlossy && evalDiffuse)
                sampleGlossy = rngNextBoolean(*e.rng, 1.0f - b.eta[1]);
            else
                sampleGlossy = evalGlossy;
            float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
            if (sampleGlossy) {
                float phi = xi0*PT_TWO_PI;
                float cosTheta = powfH(xi1, b.k[0]);
                float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta*cosTheta));
                f3 woLocal = mk3(cosfH(phi)*sinTheta, sinfH(phi)*sinTheta, cosTheta);
                e.wo = toGlobal(frameFromNormal(mk3(-e.wi.x, -e.wi.y, e.wi.z)), woLocal);
                if (e.wo.z < 0.0f)
                    return false;
                e.sampled = TGHIP_LOBE_GLOSSY_R;
            } else {
                e.wo = cosineHemisphere(xi0, xi1);
                e.sampled = TGHIP_LOBE_DIFFUSE_R;
            }
            e.pdf = pdf(s, bi, e);
            e.weight = eval(s, bi, e)/e.pdf;
            return true;
        }
        case TGHIP_BSDF_THINSHEET: {                           /* ThinSheetBsdf.cpp:49-81 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_THINSHEET))) return false;
            if (!(e.requested & TGHIP_LOBE_SPECULAR_R)) return false;
            e.wo = mk3(-e.wi.x, -e.wi.y, e.wi.z);
            e.pdf = 1.0f;
            e.sampled = TGHIP_LOBE_SPECULAR_R;
            f3 sa = ld3(b.sigma_a);
            const bool absorbing = !(sa.x == 0.0f && sa.y == 0.0f && sa.z == 0.0f);
            if (!absorbing && !b.enable_refraction) {
                e.weight = splat3(1.0f);
                return true;
            }
            float thickness = textureEval<M>(s, b.tex1, e.u, e.v).x;
            float cosThetaT;
            if (b.enable_refraction)
                e.weight = thinFilmReflectanceInterference(1.0f/b.ior, fabsf(e.wi.z), thickness*500.0f, cosThetaT);
            else
                e.weight = splat3(thinFilmReflectance(1.0f/b.ior, fabsf(e.wi.z), cosThetaT));
            f3 transmittance = splat3(1.0f) - e.weight;
            if (absorbing && cosThetaT > 0.0f)
                transmittance = transmittance*exp3((-sa)*(thickness*2.0f/cosThetaT));
            e.weight = e.weight/(1.0f - avg3(transmittance));
            return true;
        }
        case TGHIP_BSDF_OREN_NAYAR: {                          /* OrenNayarBsdf.cpp:41-59 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_OREN_NAYAR))) return false;
            if (!(e.requested & TGHIP_LOBE_DIFFUSE_R)) return false;
            if (e.wi.z <= 0.0f) return false;
            float ratio = fminf(fmaxf(bsdfRoughness<M>(s, b, e), 0.01f), 1.0f);
            const bool uniform = rngNextBoolean(*e.rng, ratio);
            float xi0 = RNG1D(*e.rng), xi1 = RNG1D(*e.rng);
            e.wo = uniform ? uniformHemisphere(xi0, xi1) : cosineHemisphere(xi0, xi1);
            e.pdf = PT_INV_TWO_PI*ratio + cosineHemispherePdf(e.wo)*(1.0f - ratio);
            e.weight = eval(s, bi, e)/e.pdf;
            e.sampled = TGHIP_LOBE_DIFFUSE_R;
            return e.wo.z > 0.0f;
        }
        case TGHIP_BSDF_ROUGH_COAT: {                          /* RoughCoatBsdf.cpp:82-159 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_ROUGH_COAT))) return false;
            if (e.wi.z <= 0.0f) return false;
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & s.bsdfs[b.sub0].lobes) != 0;
            if (!sampleR && !sampleT) return false;
            const f3 wi = e.wi;
            float eta = 1.0f/b.ior;
            float cosThetaTi;
            float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
            float substrateWeight = b.avg_transmittance*(1.0f - Fi);
            float specularProbability = Fi/(Fi + substrateWeight);
            if (sampleR && (rngNextBoolean(*e.rng, specularProbability) || !sampleT)) {
                if (!rdSampleBase<M>(e, true, false, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution)))
                    return false;
                if (sampleT) {
                    f3 brdfSpecular = e.weight*e.pdf;
                    float pdfSpecular = e.pdf*specularProbability;
                    f3 brdfSubstrate = splat3(0.0f);           /* substrateEvalAndPdf (:58-80) */
                    float pdfSubstrate = 0.0f;
                    float cosThetaTo;
                    float Fo = dielectricReflectance(eta, e.wo.z, cosThetaTo);
                    if (!(Fi == 1.0f || Fo == 1.0f)) {
                        Event q = e;
                        q.wi = mk3(wi.x*eta, wi.y*eta, copysignf(cosThetaTi, wi.z));
                        q.wo = mk3(e.wo.x*eta, e.wo.y*eta, copysignf(cosThetaTo, e.wo.z));
                        pdfSubstrate = Next::pdf(s, b.sub0, q);
                        pdfSubstrate *= eta*eta*fabsf(e.wo.z/cosThetaTo);
                        float compressionProjection = eta*eta*e.wo.z/cosThetaTo;
                        f3 substrateF = absorb(b, Next::eval(s, b.sub0, q), cosThetaTo, cosThetaTi);
                        brdfSubstrate = substrateF*(compressionProjection*(1.0f - Fi)*(1.0f - Fo));
                    }
                    pdfSubstrate *= 1.0f - specularProbability;
                    e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
                    e.pdf = pdfSpecular + pdfSubstrate;
                }
                return true;
            }
            const f3 wiSubstrate = mk3(wi.x*eta, wi.y*eta, cosThetaTi);
            e.wi = wiSubstrate;
            bool success = Next::sample(s, b.sub0, e);
            e.wi = wi;
            if (!success) return false;
            float cosThetaTo;
            float Fo = dielectricReflectance(b.ior, e.wo.z, cosThetaTo);
            if (Fo == 1.0f) return false;
            float cosThetaSubstrate = e.wo.z;
            e.wo = mk3(e.wo.x*b.ior, e.wo.y*b.ior, cosThetaTo);
            e.weight = e.weight*((1.0f - Fi)*(1.0f - Fo));
            e.weight = absorb(b, e.weight, cosThetaSubstrate, cosThetaTi);
            e.weight = e.weight*(wi.z/wiSubstrate.z);
            e.pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
            if (sampleR) {
                f3 brdfSubstrate = e.weight*e.pdf;
                float pdfSubstrate = e.pdf*(1.0f - specularProbability);
                float r = bsdfRoughness<M>(s, b, e);
                f3 brdfSpecular = rdEvalBase(e, true, false, r, b.ior, mfDist<M>(b.distribution));
                float pdfSpecular = rdPdfBase(e, true, false, r, b.ior, mfDist<M>(b.distribution));
                pdfSpecular *= specularProbability;
                e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
                e.pdf = pdfSpecular + pdfSubstrate;
            }
            return true;
        }
        case TGHIP_BSDF_TRANSPARENCY:                          /* TransparencyBsdf.cpp:43-46 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_TRANSPARENCY)))) return false;
            return Next::sample(s, b.sub0, e);
        default:                                               /* null, forward */
            return false;
        }
    }

    static __device__ float pdf(const DeviceScene &s, int bi, const Event &e)
    {
        const TgHipBsdf &b = s.bsdfs[bi];
        switch (b.type) {
        case TGHIP_BSDF_LAMBERT: case TGHIP_BSDF_ERROR:        /* LambertBsdf.cpp:61-68 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_LAMBERT) | BSDF_BIT(TGHIP_BSDF_ERROR)))) return 0.0f;
            if (!(e.requested & TGHIP_LOBE_DIFFUSE_R)) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            return cosineHemispherePdf(e.wo);
        case TGHIP_BSDF_MIRROR: case TGHIP_BSDF_CONDUCTOR:
            if (!(M & (BSDF_BIT(TGHIP_BSDF_MIRROR) | BSDF_BIT(TGHIP_BSDF_CONDUCTOR)))) return 0.0f;
            return ((e.requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e.wi, e.wo)) ? 1.0f : 0.0f;
        case TGHIP_BSDF_ROUGH_CONDUCTOR: {                     /* RoughConductorBsdf.cpp:127-143 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_CONDUCTOR)))) return 0.0f;
            if (!(e.requested & TGHIP_LOBE_GLOSSY_R)) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            float sampleAlpha = mfRoughnessToAlpha(mfDist<M>(b.distribution), bsdfRoughness<M>(s, b, e));
            f3 hr = normalized(e.wi + e.wo);
            return mfPdf(mfDist<M>(b.distribution), sampleAlpha, hr)*0.25f/dot(e.wi, hr);
        }
        case TGHIP_BSDF_SMOOTH_COAT: {                         /* SmoothCoatBsdf.cpp:179-214 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_SMOOTH_COAT)))) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            bool sampleR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool sampleT = (e.requested & s.bsdfs[b.sub0].lobes) != 0;
            float eta = 1.0f/b.ior;
            float cosThetaTi, cosThetaTo;
            float Fi = dielectricReflectance(eta, e.wi.z, cosThetaTi);
            dielectricReflectance(eta, e.wo.z, cosThetaTo);
            Event q = e;
            q.wi = mk3(e.wi.x*eta, e.wi.y*eta, copysignf(cosThetaTi, e.wi.z));
            q.wo = mk3(e.wo.x*eta, e.wo.y*eta, copysignf(cosThetaTo, e.wo.z));
            if (sampleR && sampleT) {
                float substrateWeight = b.avg_transmittance*(1.0f - Fi);
                float specularProbability = Fi/(Fi + substrateWeight);
                if (checkReflectionConstraint(e.wi, e.wo))
                    return specularProbability;
                return Next::pdf(s, b.sub0, q)*(1.0f - specularProbability)*eta*eta*fabsf(e.wo.z/cosThetaTo);
            } else if (sampleT) {
                return Next::pdf(s, b.sub0, q)*eta*eta*fabsf(e.wo.z/cosThetaTo);
            } else if (sampleR) {
                return checkReflectionConstraint(e.wi, e.wo) ? 1.0f : 0.0f;
            }
            return 0.0f;
        }
        case TGHIP_BSDF_DIELECTRIC: {                          /* DielectricBsdf.cpp:143-164 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_DIELECTRIC)))) return 0.0f;
            bool sampleR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_SPECULAR_T) && b.enable_refraction;
            float eta = e.wi.z < 0.0f ? b.ior : 1.0f/b.ior;
            float cosThetaT = 0.0f;
            float F = dielectricReflectance(eta, fabsf(e.wi.z), cosThetaT);
            if (e.wi.z*e.wo.z >= 0.0f) {
                if (sampleR && checkReflectionConstraint(e.wi, e.wo)) return sampleT ? F : 1.0f;
                return 0.0f;
            }
            if (sampleT && checkRefractionConstraint(e.wi, e.wo, eta, cosThetaT)) return sampleR ? 1.0f - F : 1.0f;
            return 0.0f;
        }
        case TGHIP_BSDF_ROUGH_DIELECTRIC: {
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_DIELECTRIC)))) return 0.0f;
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_GLOSSY_T) && b.enable_refraction;
            return rdPdfBase(e, sampleR, sampleT, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution));
        }
        case TGHIP_BSDF_PLASTIC: {                             /* PlasticBsdf.cpp:153-177 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_PLASTIC)))) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            bool sampleR = (e.requested & TGHIP_LOBE_SPECULAR_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (sampleR && sampleT) {
                float Fi = dielectricReflectance(1.0f/b.ior, e.wi.z);
                float substrateWeight = b.avg_transmittance*(1.0f - Fi);
                float specularProbability = Fi/(Fi + substrateWeight);
                if (checkReflectionConstraint(e.wi, e.wo)) return specularProbability;
                return cosineHemispherePdf(e.wo)*(1.0f - specularProbability);
            } else if (sampleT) {
                return cosineHemispherePdf(e.wo);
            } else if (sampleR) {
                return checkReflectionConstraint(e.wi, e.wo) ? 1.0f : 0.0f;
            }
            return 0.0f;
        }
        case TGHIP_BSDF_ROUGH_PLASTIC: {                       /* RoughPlasticBsdf.cpp:185-213 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_ROUGH_PLASTIC)))) return 0.0f;
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (!sampleR && !sampleT) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            float glossyPdf = sampleR ? rdPdfBase(e, true, false, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution)) : 0.0f;
            float diffusePdf = sampleT ? cosineHemispherePdf(e.wo) : 0.0f;
            if (sampleT && sampleR) {
                float Fi = dielectricReflectance(1.0f/b.ior, e.wi.z);
                float substrateW = avg3(ld3(s.textures[b.albedo].avg));
                float substrateWeight = substrateW*b.avg_transmittance*(1.0f - Fi);
                float specularProbability = Fi/(Fi + substrateWeight);
                diffusePdf *= (1.0f - specularProbability);
                glossyPdf *= specularProbability;
            }
            return glossyPdf + diffusePdf;
        }
        case TGHIP_BSDF_MIXED: {                               /* MixedBsdf.cpp:124-130 */
            if (!(M & (BSDF_BIT(TGHIP_BSDF_MIXED)))) return 0.0f;
            float ratio;
            if (!mixedRatio(s, b, e, ratio)) return 0.0f;
            return Next::pdf(s, b.sub0, e)*ratio + Next::pdf(s, b.sub1, e)*(1.0f - ratio);
        }
        case TGHIP_BSDF_DIFFUSE_TRANSMISSION: {                /* DiffuseTransmissionBsdf.cpp:77-88 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_DIFFUSE_TRANSMISSION))) return 0.0f;
            bool sampleR = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0, sampleT = (e.requested & TGHIP_LOBE_DIFFUSE_T) != 0;
            if (!sampleR && !sampleT) return 0.0f;
            float transmittanceProbability = sampleR && sampleT ? b.eta[0] : (sampleR ? 0.0f : 1.0f);
            float factor = e.wi.z*e.wo.z < 0.0f ? transmittanceProbability : 1.0f - transmittanceProbability;
            return factor*cosineHemispherePdf(e.wo);
        }
        case TGHIP_BSDF_PHONG: {                               /* PhongBsdf.cpp:101-124 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_PHONG))) return 0.0f;
            bool evalGlossy = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0, evalDiffuse = (e.requested & TGHIP_LOBE_DIFFUSE_R) != 0;
            if (!evalGlossy && !evalDiffuse) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            float result = 0.0f;
            if (evalGlossy) {
                float cosTheta = dot(mk3(-e.wi.x, -e.wi.y, e.wi.z), e.wo);
                if (cosTheta > 0.0f)
                    result += powfH(cosTheta, b.eta[0])*b.k[1];
            }
            if (evalDiffuse && evalGlossy)
                result = result*(1.0f - b.eta[1]) + b.eta[1]*cosineHemispherePdf(e.wo);
            else if (evalDiffuse)
                result = cosineHemispherePdf(e.wo);
            return result;
        }
        case TGHIP_BSDF_THINSHEET:                             /* ThinSheetBsdf.cpp:112-119 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_THINSHEET))) return 0.0f;
            return ((e.requested & TGHIP_LOBE_SPECULAR_R) && checkReflectionConstraint(e.wi, e.wo)) ? 1.0f : 0.0f;
        case TGHIP_BSDF_OREN_NAYAR: {                          /* OrenNayarBsdf.cpp:125-135 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_OREN_NAYAR))) return 0.0f;
            if (!(e.requested & TGHIP_LOBE_DIFFUSE_R)) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            float ratio = fminf(fmaxf(bsdfRoughness<M>(s, b, e), 0.01f), 1.0f);
            return PT_INV_TWO_PI*ratio + cosineHemispherePdf(e.wo)*(1.0f - ratio);
        }
        case TGHIP_BSDF_ROUGH_COAT: {                          /* RoughCoatBsdf.cpp:259-298 */
            if (!(M & BSDF_BIT(TGHIP_BSDF_ROUGH_COAT))) return 0.0f;
            bool sampleR = (e.requested & TGHIP_LOBE_GLOSSY_R) != 0;
            bool sampleT = (e.requested & s.bsdfs[b.sub0].lobes) != 0;
            if (!sampleT && !sampleR) return 0.0f;
            if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
            float eta = 1.0f/b.ior;
            float cosThetaTi, cosThetaTo;
            float Fi = dielectricReflectance(eta, e.wi.z, cosThetaTi);
            float Fo = dielectricReflectance(eta, e.wo.z, cosThetaTo);
            float specularProbability;
            if (sampleR && sampleT) {
                float substrateWeight = b.avg_transmittance*(1.0f - Fi);
                specularProbability = Fi/(Fi + substrateWeight);
            } else {
                specularProbability = sampleR ? 1.0f : 0.0f;
            }
            float glossyPdf = 0.0f;
            if (sampleR)
                glossyPdf = rdPdfBase(e, true, false, bsdfRoughness<M>(s, b, e), b.ior, mfDist<M>(b.distribution));
            float substratePdf = 0.0f;
            if (sampleT && Fi < 1.0f && Fo < 1.0f) {
                Event q = e;
                q.wi = mk3(e.wi.x*eta, e.wi.y*eta, copysignf(cosThetaTi, e.wi.z));
                q.wo = mk3(e.wo.x*eta, e.wo.y*eta, copysignf(cosThetaTo, e.wo.z));
                substratePdf = Next::pdf(s, b.sub0, q);
                substratePdf *= eta*eta*fabsf(e.wo.z/cosThetaTo);
            }
            return glossyPdf*specularProbability + substratePdf*(1.0f - specularProbability);
        }
        case TGHIP_BSDF_TRANSPARENCY:
            if (!(M & (BSDF_BIT(TGHIP_BSDF_TRANSPARENCY)))) return 0.0f;
            return Next::pdf(s, b.sub0, e);
        default:
            return 0.0f;
        }
    }
};

// nesting deeper than PT_MAX_BSDF_DEPTH is rejected at upload time; terminate the template chain
template<uint32_t M> struct BsdfOps<PT_MAX_BSDF_DEPTH, M> {
    static __device__ f3 eval(const DeviceScene &, int, const Event &) { return splat3(0.0f); }
    static __device__ bool sample(const DeviceScene &, int, Event &) { return false; }
    static __device__ float pdf(const DeviceScene &, int, const Event &) { return 0.0f; }
};

/* Bsdf::eta (Bsdf.hpp:99-103; DielectricBsdf.cpp:166-174, RoughDielectricBsdf.cpp:274-280) */
PT_DEV float bsdfEta(const DeviceScene &s, int bi, const Event &e)
{
    const TgHipBsdf &b = s.bsdfs[bi];
    if (b.type == TGHIP_BSDF_DIELECTRIC || b.type == TGHIP_BSDF_ROUGH_DIELECTRIC) {
        if (e.wi.z*e.wo.z >= 0.0f) return 1.0f;
        return e.wi.z < 0.0f ? b.ior : 1.0f/b.ior;
    }
    return 1.0f;
}
/* radiance-transport wrappers (adjoint == false, Bsdf.hpp:71-97) */
template<uint32_t M>
PT_DEV f3 bsdfEval(const DeviceScene &s, int bi, const Event &e)
{
    f3 f = BsdfOps<0, M>::eval(s, bi, e);
    if (M & (BSDF_BIT(TGHIP_BSDF_DIELECTRIC) | BSDF_BIT(TGHIP_BSDF_ROUGH_DIELECTRIC))) f = f*sqr(bsdfEta(s, bi, e));
    return f;
}
template<uint32_t M>
PT_DEV float bsdfPdf(const DeviceScene &s, int bi, const Event &e) { return BsdfOps<0, M>::pdf(s, bi, e); }
template<uint32_t M>
PT_DEV bool bsdfSample(const DeviceScene &s, int bi, Event &e)
{
    if (!BsdfOps<0, M>::sample(s, bi, e)) return false;
    if (M & (BSDF_BIT(TGHIP_BSDF_DIELECTRIC) | BSDF_BIT(TGHIP_BSDF_ROUGH_DIELECTRIC))) e.weight = e.weight*sqr(bsdfEta(s, bi, e));
    return true;
}

// ---------------------------------------------------------------------------------------------
// Primitive tests on 48-byte records (shared by the traversal kernels and the analytic light hit)
// ---------------------------------------------------------------------------------------------
struct RayD { f3 o, d; float tmin, tmax; };

/* PROVENANCE of the next ~80 lines (dotEmbree, rcpEmbree, the slab test, triangleEmbree) and of pt_kernels.h's flatOrderedWalk: they restate the
 * ARITHMETIC of Embree 2.11 as the reference vendors it (thirdparty/embree/kernels/geometry/triangle_intersector_moeller.h,
 * kernels/bvh/bvh_intersector_node.h, bvh_traverser1.h, common/simd/sse.h: rcp) operation by operation, because the reference's hit distances
 * and visiting order are functions of exactly those roundings.  Embree: Copyright 2009-2016 Intel Corporation, Apache License 2.0
 * (http://www.apache.org/licenses/LICENSE-2.0); derived work distributed under the same terms, "AS IS", without warranties of any kind. */
/* Embree's arithmetic in its triangle test, bit for bit (oracle/oracle.c: intel_rcpps / embree_rcp / edot has the derivation): the dot product
 * associates from the right (common/math/vec3.h:182), and t / u / v are PRODUCTS with rcp(absDen) = r*(2 - r*a), r = RCPPS(a) -- the Intel
 * instruction's estimate: 2^25/(4097 + 2 i) rounded to an integer for the operand's top 11 mantissa bits i, the operand's exponent negated
 * (simd/vfloat4_sse2.h:166-173).  The integer quotient comes from v_rcp_f32 and is corrected by the exact remainder, so no division is
 * executed: cheaper than the three exact divisions this replaced.  tests/test_gpu_libm.py holds embreeRcp to the oracle's. */
PT_DEV float dotEmbree(f3 a, f3 b) { return a.x*b.x + (a.y*b.y + a.z*b.z); }
PT_DEV float rcppsIntel(float x)
{
    const uint32_t u = __float_as_uint(x), sign = u & 0x80000000u, e = (u >> 23) & 0xffu, i = (u >> 12) & 0x7ffu;
    const uint32_t d = 4097u + 2u*i;
    int q = (int)(33554432.0f*__builtin_amdgcn_rcpf((float)d));          // within one of the quotient's floor
    int r = (int)(33554432u - (uint32_t)q*d);
    if (r < 0) { q -= 1; r += (int)d; }
    if (r >= (int)d) { q += 1; r -= (int)d; }
    if (2*r > (int)d) q += 1;                                             // d is odd: no ties
    uint32_t bits = sign | ((253u - e) << 23) | ((uint32_t)(q - 4096) << 11);
    if (e >= 253u) bits = (e == 255u && (u & 0x7fffffu)) ? (u | 0x00400000u) : sign;   // denormal result / 1/inf: zero; NaN quieted
    if (e == 0u) bits = sign | 0x7f800000u;                                // zeros and denormals: infinity
    return __uint_as_float(bits);
}
PT_DEV float embreeRcp(float a) { const float r = rcppsIntel(a); return r*(2.0f - r*a); }
// Embree's slab test of one child box of a BVH4 node as its SSE4.2 single-ray traversal makes it (kernels/bvh/bvh_intersector_node.h:162-195,
// TravRay :30-47): rdir = rcp(zero_fix(dir)), planes (bound - org)*rdir, near / far by the sign of rdir, maxi / mini and the final comparison on
// the floats' bit patterns as signed integers.  The user-geometry BVH has one primitive per leaf, so a primitive's own box is the box a ray
// must pass to reach it (oracle.c: embree_box_visible; used for the light a shadow ray is aimed at, pt_wavefront.h: k_trace_shadow<., FORWARD>).
PT_DEV bool embreeIntGreater(float a, float b) { return __float_as_int(a) > __float_as_int(b); }
PT_DEV bool embreeBoxVisible(f3 o, f3 d, float tmin, float tmax, f3 lo, f3 hi)
{
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    const float tNear = fmaxf(tmin, 0.0f), tFar = fmaxf(tmax, 0.0f);
    float n[3], f[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = fabsf(dd[k]) < 1e-18f ? 1e-18f : dd[k];
        const float rdir = embreeRcp(a);
        n[k] = ((rdir >= 0.0f ? l[k] : h[k]) - oo[k])*rdir;
        f[k] = ((rdir >= 0.0f ? h[k] : l[k]) - oo[k])*rdir;
    }
    const float nxy = embreeIntGreater(n[0], n[1]) ? n[0] : n[1], nzt = embreeIntGreater(n[2], tNear) ? n[2] : tNear;
    const float nearT = embreeIntGreater(nxy, nzt) ? nxy : nzt;
    const float fxy = embreeIntGreater(f[0], f[1]) ? f[1] : f[0], fzt = embreeIntGreater(f[2], tFar) ? tFar : f[2];
    const float farT = embreeIntGreater(fxy, fzt) ? fzt : fxy;
    return !embreeIntGreater(nearT, farT);
}

/* Embree MoellerTrumboreIntersector1 (thirdparty/embree/kernels/geometry/triangle_intersector_moeller.h:76-113, finalize() :43-49);
 * Embree's e1 = v0 - v1 = -rec.b, e2 = v2 - v0 = rec.c. */
#ifndef PT_TRI_BRANCHLESS
#define PT_TRI_BRANCHLESS 0    /* measured, profiles/r6_ab_lane_cliff.txt: level on the metric's workload, the closest-hit walk of mesh1m 528 -> 537 us */
#endif
PT_DEV bool triTest(f3 v0, f3 b, f3 c, const RayD &ray, float tmax, float &t, float &u, float &v)
{
    f3 e1 = -b, e2 = c;
    f3 Ng = cross(e1, e2);
    f3 C = v0 - ray.o;
    f3 R = cross(ray.d, C);
    float den = dotEmbree(Ng, ray.d);
    float absDen = fabsf(den);
    float sgn = den < 0.0f ? -1.0f : 1.0f;
    float U = dotEmbree(R, e2)*sgn;
    float V = dotEmbree(R, e1)*sgn;
#if PT_TRI_BRANCHLESS
    // Round 6: no early exits.  A wave64 VALU instruction with eight or fewer enabled lanes issues at a QUARTER of the rate (tools/ubench_lanes.hip,
    // profiles/r6_ubench_lane_masks.txt: 935 G wave-instructions/s with 9 .. 64 lanes enabled, 241 with 1 .. 8, whichever lanes they are), and that is
    // where the two exits put the rest of the test: of the ~20 lanes of a walk's wave that test a record in a turn, a handful pass the edge
    // functions and two find a hit (bench.py: roofline.valu.walk), so the distance test and the reciprocal -- some 40 instructions -- ran in nearly
    // every turn for one to eight lanes, at four times their price.  Every lane with a record computes them now; the values are the same.
    const bool inside = den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absDen;
    const float T = dotEmbree(Ng, C)*sgn;
    const bool inRange = T > absDen*ray.tmin && T < absDen*tmax;
    const float rcpAll = embreeRcp(absDen);
    t = T*rcpAll; u = U*rcpAll; v = V*rcpAll;
    return inside && inRange;
#else
    if (!(den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absDen))
        return false;
    float T = dotEmbree(Ng, C)*sgn;
    if (!(T > absDen*ray.tmin && T < absDen*tmax))
        return false;
#endif
#ifdef PT_TRI_DIVIDE   /* experiment (profiles/README.md): the exact divisions rounds 1-3 ran here, to price Embree's reciprocal against them */
    t = T/absDen; u = U/absDen; v = V/absDen;
#else
    const float rcpAbsDen = embreeRcp(absDen);
    t = T*rcpAbsDen; u = U*rcpAbsDen; v = V*rcpAbsDen;
#endif
    return true;
}

/* Quad::intersect (Quad.cpp:71-98) */
PT_DEV bool quadTest(f3 base, f3 edge0, f3 edge1, float invUvSq0, float invUvSq1, f3 n, const RayD &ray, float tmax,
                     float &t, float &l0, float &l1)
{
    float nDotW = dot(ray.d, n);
    if (fabsf(nDotW) < 1e-6f)
        return false;
    float tt = dot(n, base - ray.o)/nDotW;
    if (tt < ray.tmin || tt > tmax)
        return false;
    f3 q = ray.o + ray.d*tt;
    f3 v = q - base;
    float a = dot(v, edge0)*invUvSq0;
    float b = dot(v, edge1)*invUvSq1;
    if (a < 0.0f || a > 1.0f || b < 0.0f || b > 1.0f)
        return false;
    t = tt; l0 = a; l1 = b;
    return true;
}

/* Cube::intersect (Cube.cpp:94-125) */
template<typename OP>   // OP: pointer to TgHipObject (generic or constant address space)
PT_DEV bool cubeTest(OP op, const RayD &ray, float tmax, float &t, bool &backSide)
{
    const auto &o = *op;
    f3 p = mat3TMul(o.rot, ray.o - ld3(o.pos));
    f3 d = mat3TMul(o.rot, ray.d);
    float pa[3] = {p.x, p.y, p.z}, da[3] = {d.x, d.y, d.z};
    float ttMin = ray.tmin, ttMax = tmax;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float invD = 1.0f/da[i];
        float relMin = -o.scale[i] - pa[i];
        float relMax = o.scale[i] - pa[i];
        if (invD >= 0.0f) {
            ttMin = fmaxf(ttMin, relMin*invD);
            ttMax = fminf(ttMax, relMax*invD);
        } else {
            ttMax = fminf(ttMax, relMin*invD);
            ttMin = fmaxf(ttMin, relMax*invD);
        }
    }
    if (ttMin <= ttMax) {
        if (ttMin > ray.tmin && ttMin < tmax) { t = ttMin; backSide = false; return true; }
        if (ttMax > ray.tmin && ttMax < tmax) { t = ttMax; backSide = true; return true; }
    }
    return false;
}

/* Sphere::intersect (Sphere.cpp:69-94); radius in scale[0] */
template<typename OP>
PT_DEV bool sphereTest(OP op, const RayD &ray, float tmax, float &t, bool &backSide)
{
    const auto &o = *op;
    f3 p = ray.o - ld3(o.pos);
    float B = dot(p, ray.d);
    float C = lengthSq(p) - o.scale[0]*o.scale[0];
    float detSq = B*B - C;
    if (detSq >= 0.0f) {
        float det = sqrtf(detSq);
        float tt = -B - det;
        if (tt < tmax && tt > ray.tmin) { t = tt; backSide = false; return true; }
        tt = -B + det;
        if (tt < tmax && tt > ray.tmin) { t = tt; backSide = true; return true; }
    }
    return false;
}

/* Disk::intersect (Disk.cpp:63-85): pos = _center, normal = _n, scale = {_r, _cosApex}; backSide = outside the emission cone */
template<typename OP>
PT_DEV bool diskTest(OP op, const RayD &ray, float tmax, float &t, float &rSq, bool &backSide)
{
    const auto &o = *op;
    f3 n = ld3(o.normal), center = ld3(o.pos);
    float nDotW = dot(ray.d, n);
    float tt = dot(n, center - ray.o)/nDotW;
    if (tt < ray.tmin || tt > tmax)
        return false;
    f3 v = (ray.o + ray.d*tt) - center;
    float r2 = lengthSq(v);
    if (r2 > o.scale[0]*o.scale[0])
        return false;
    t = tt; rSq = r2; backSide = -nDotW < o.scale[1];
    return true;
}
/* Disk::intersectionInfo (Disk.cpp:114-129): uv of the hit point; edge0 = tangent, edge1 = bitangent */
PT_DEV void diskSurface(const TgHipObject &o, f3 hp, float rSq, float &u, float &v)
{
    f3 d = hp - ld3(o.pos);
    float x = dot(d, ld3(o.edge1)), y = dot(d, ld3(o.edge0));
    v = sqrtf(rSq)/o.scale[0];
    u = (x == 0.0f && y == 0.0f) ? 0.0f : (atan2fH(y, x)*PT_INV_TWO_PI + 0.5f);
}

/* Cylinder::intersect (Cylinder.cpp:55-108): pos = _pos, rot = _rot, scale = {_radius, _halfHeight, _capped}; cap = +-1 when a
 * cap was hit (its sign), 0 for the side */
template<typename OP>
PT_DEV bool cylinderTest(OP op, const RayD &ray, float tmax, float &tOut, bool &backSide, float &cap)
{
    const auto &o = *op;
    const float radius = o.scale[0], halfHeight = o.scale[1], invRadius = 1.0f/radius;
    f3 pLocal = mat3TMul(o.rot, ray.o - ld3(o.pos));
    f3 dLocal = mat3TMul(o.rot, ray.d);
    float px = pLocal.x*invRadius, pz = pLocal.z*invRadius, dx = dLocal.x*invRadius, dz = dLocal.z*invRadius;
    bool didHit = false;
    float farT = tmax;
    if (o.scale[2] != 0.0f && fabsf(dLocal.y) > 1e-6f) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float sign = k == 0 ? 1.0f : -1.0f;
            float t = (sign*halfHeight - pLocal.y)/dLocal.y;
            if (t > ray.tmin && t < farT) {
                float hx = px + t*dx, hz = pz + t*dz;
                if (hx*hx + hz*hz < 1.0f) { didHit = true; cap = sign; backSide = sign*dLocal.y > 0.0f; farT = t; }
            }
        }
    }
    float A = dx*dx + dz*dz, B = px*dx + pz*dz, C = px*px + pz*pz - 1.0f;
    float detSq = B*B - A*C;
    if (detSq >= 0.0f) {
        float det = sqrtf(detSq);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float sign = k == 0 ? 1.0f : -1.0f;
            float t = (-B - sign*det)/A;
            if (t > ray.tmin && t < farT) {
                float h = pLocal.y + dLocal.y*t;
                if (h >= -halfHeight && h <= halfHeight) { didHit = true; cap = 0.0f; backSide = sign < 0.0f; farT = t; }
            }
        }
    }
    if (didHit) tOut = farT;
    return didHit;
}
/* Cylinder::intersectionInfo (Cylinder.cpp:122-132).  The reference keeps what Cylinder::intersect computed in the cylinder's own space --
   pHit = p + t d in the unit-radius cross-section, h = pLocal.y + dLocal.y t (:70-74, :92-97) -- so the normal and the uv are functions of the
   RAY and t (the same expressions as in cylinderTest above), not of the world-space hit point: oracle.c: cylinder_surface. */
PT_DEV void cylinderSurface(const TgHipObject &o, const RayD &ray, float t, float cap, f3 &n, float &u, float &v)
{
    const float invRadius = 1.0f/o.scale[0];
    const f3 pLocal = mat3TMul(o.rot, ray.o - ld3(o.pos));
    const f3 dLocal = mat3TMul(o.rot, ray.d);
    const float px = pLocal.x*invRadius, pz = pLocal.z*invRadius, dx = dLocal.x*invRadius, dz = dLocal.z*invRadius;
    const float hx = px + t*dx, hz = pz + t*dz;
    if (cap != 0.0f) {
        n = mat3Mul(o.rot, mk3(0.0f, cap, 0.0f));
        u = hx*0.5f + 0.5f; v = hz*0.5f + 0.5f;
    } else {
        const float h = pLocal.y + dLocal.y*t;
        n = mat3Mul(o.rot, mk3(hx, 0.0f, hz));
        u = atan2fH(hz, hx)*PT_INV_TWO_PI + 0.5f;
        v = h*(0.5f/o.scale[1]) + 0.5f;
    }
}

/* normal and uv of a point on a cube / sphere (Cube.cpp:157-170, Sphere.cpp:120-129) */
PT_DEV void cubeSurface(const TgHipObject &o, f3 hp, f3 &n, float &u, float &v)
{
    f3 p = mat3TMul(o.rot, hp - ld3(o.pos));
    float pa[3] = {p.x, p.y, p.z};
    float ex[3] = {fabsf(p.x) - o.scale[0], fabsf(p.y) - o.scale[1], fabsf(p.z) - o.scale[2]};
    int dim = ex[0] > ex[1] ? (ex[0] > ex[2] ? 0 : 2) : (ex[1] > ex[2] ? 1 : 2);
    float nn[3] = {0.0f, 0.0f, 0.0f};
    nn[dim] = pa[dim] < 0.0f ? -1.0f : 1.0f;
    float uvw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) uvw[i] = (pa[i]/o.scale[i])*0.5f + 0.5f;
    n = mat3Mul(o.rot, mk3(nn[0], nn[1], nn[2]));
    u = uvw[(dim + 1) % 3]; v = uvw[(dim + 2) % 3];
}
PT_DEV void sphereSurface(const TgHipObject &o, f3 hp, f3 &n, float &u, float &v)
{
    n = (hp - ld3(o.pos))/o.scale[0];
    f3 localN = mat3TMul(o.rot, n);
    u = atan2fH(localN.y, localN.x)*PT_INV_TWO_PI + 0.5f;
    v = acosfExact(clampf(localN.z, -1.0f, 1.0f))*PT_INV_PI;
    if (isnan(u)) u = 0.0f;
}

/* Tests record `ri` against the ray; on a hit updates (tmax, hit) -- the leaf body of both traversal kernels.
 * UNIFORM: `ri` is the same for every lane (flat-list traversal), so the record and its object are fetched
 * through the constant address space, i.e. with scalar loads. */
// KINDS: the record kinds the scene (or the kernel variant) can contain; tests of absent kinds fold away, which keeps
// the hot traversal kernels of triangle scenes at their register budget whatever analytic primitives exist elsewhere.
#define KIND_BIT(k)   (1u << (k))
#define KINDS_ALL     0x7Fu
#define KINDS_MESH    (KIND_BIT(TGHIP_REC_TRIANGLE) | KIND_BIT(TGHIP_REC_QUAD))   /* triangle meshes + quads (materialtest) */
template<bool UNIFORM, uint32_t KINDS = KINDS_ALL>
PT_DEV bool testRecordLoaded(const DeviceScene &s, uint32_t ri, float4 r0, float4 r1, float4 r2, const RayD &ray, float &tmax, float4 &hit, uint32_t &hitMeta)
{
    uint32_t meta = __float_as_uint(r0.w);
    uint32_t kind = TGHIP_REC_KIND(meta);
    float t, u = 0.0f, v = 0.0f;
    bool ok;
    if ((KINDS & KIND_BIT(TGHIP_REC_TRIANGLE)) && (kind == TGHIP_REC_TRIANGLE || KINDS == KIND_BIT(TGHIP_REC_TRIANGLE))) {
        ok = triTest(xyz(r0), xyz(r1), xyz(r2), ray, tmax, t, u, v);
    } else if ((KINDS & KIND_BIT(TGHIP_REC_QUAD)) && kind == TGHIP_REC_QUAD) {
        f3 n = UNIFORM ? ld3(asConst(s.objects)[TGHIP_REC_OBJECT(meta)].normal) : ld3(s.objects[TGHIP_REC_OBJECT(meta)].normal);
        ok = quadTest(xyz(r0), xyz(r1), xyz(r2), r1.w, r2.w, n, ray, tmax, t, u, v);
    } else if ((KINDS & KIND_BIT(TGHIP_REC_CUBE)) && kind == TGHIP_REC_CUBE) {
        bool back;
        if (UNIFORM) ok = cubeTest(asConst(s.objects) + TGHIP_REC_OBJECT(meta), ray, tmax, t, back);
        else         ok = cubeTest(s.objects + TGHIP_REC_OBJECT(meta), ray, tmax, t, back);
        u = back ? 1.0f : 0.0f;
    } else if ((KINDS & KIND_BIT(TGHIP_REC_DISK)) && kind == TGHIP_REC_DISK) {   /* v carries the squared distance from the centre */
        bool back;
        if (UNIFORM) ok = diskTest(asConst(s.objects) + TGHIP_REC_OBJECT(meta), ray, tmax, t, v, back);
        else         ok = diskTest(s.objects + TGHIP_REC_OBJECT(meta), ray, tmax, t, v, back);
        u = back ? 1.0f : 0.0f;
    } else if ((KINDS & KIND_BIT(TGHIP_REC_CYLINDER)) && kind == TGHIP_REC_CYLINDER) {   /* v carries the cap sign */
        bool back;
        if (UNIFORM) ok = cylinderTest(asConst(s.objects) + TGHIP_REC_OBJECT(meta), ray, tmax, t, back, v);
        else         ok = cylinderTest(s.objects + TGHIP_REC_OBJECT(meta), ray, tmax, t, back, v);
        u = back ? 1.0f : 0.0f;
    } else if ((KINDS & KIND_BIT(TGHIP_REC_SPHERE)) && kind == TGHIP_REC_SPHERE) {
        bool back;
        if (UNIFORM) ok = sphereTest(asConst(s.objects) + TGHIP_REC_OBJECT(meta), ray, tmax, t, back);
        else         ok = sphereTest(s.objects + TGHIP_REC_OBJECT(meta), ray, tmax, t, back);
        u = back ? 1.0f : 0.0f;
    } else {
        ok = false;                                    /* instance records are entered, not tested (traverseClosestInst) */
    }
#if PT_TRI_BRANCHLESS
    if (!UNIFORM) {                                    /* (selects, not a region a few lanes enter: triTest above says why) */
        tmax = ok ? t : tmax;
        hit = make_float4(ok ? t : hit.x, ok ? u : hit.y, ok ? v : hit.z, ok ? __int_as_float((int)ri) : hit.w);
        hitMeta = meta;                                /* (read by the callers only behind `ok`) */
        return ok;
    }
#endif
    if (ok) {
        tmax = t;
        hit = make_float4(t, u, v, __int_as_float((int)ri));
        hitMeta = meta;
    }
    return ok;
}
template<bool UNIFORM, uint32_t KINDS = KINDS_ALL>
PT_DEV bool testRecord(const DeviceScene &s, uint32_t ri, const RayD &ray, float &tmax, float4 &hit, uint32_t &hitMeta)
{
    float4 r0, r1, r2;
    if (UNIFORM) {
        const PT_CONST_AS float *rp = asConst(reinterpret_cast<const float *>(s.recs)) + ri*12;
        r0 = make_float4(rp[0], rp[1], rp[2], rp[3]);
        r1 = make_float4(rp[4], rp[5], rp[6], rp[7]);
        r2 = make_float4(rp[8], rp[9], rp[10], rp[11]);
    } else {
        r0 = ld4(s.recs, ri*3u + 0u); r1 = ld4(s.recs, ri*3u + 1u); r2 = ld4(s.recs, ri*3u + 2u);
    }
    return testRecordLoaded<UNIFORM, KINDS>(s, ri, r0, r1, r2, ray, tmax, hit, hitMeta);
}
template<bool UNIFORM, uint32_t KINDS = KINDS_ALL>
PT_DEV void testRecord(const DeviceScene &s, uint32_t ri, const RayD &ray, float &tmax, float4 &hit)
{
    uint32_t meta;
    (void)testRecord<UNIFORM, KINDS>(s, ri, ray, tmax, hit, meta);
}

// ---------------------------------------------------------------------------------------------
// IntersectionInfo (primitives/IntersectionInfo.hpp:11-22)
// ---------------------------------------------------------------------------------------------
struct Info {
    f3 Ng, Ns, p;
    float u, v;
    int object, bsdf;
    bool backSide;
    // FEAT_BUMP variants only: Primitive::tangentSpace of the primitive that was hit (TriangleMesh.cpp:362-384, Quad.cpp:133-139, Cube.cpp:172-182,
    // Sphere.cpp:131-137, Disk.cpp:129-140, Cylinder.cpp:135-141; an `instances` primitive has none, Instance.cpp:348-351)
    f3 T, B;
    bool hasTB;
};

// Quaternion<float>::operator*(Vec3) (math/Quaternion.hpp:78-88); q = (w, x, y, z)
PT_DEV f3 quatRotate(float qw, f3 q, f3 o)
{
    float tx = 2.0f*(q.y*o.z - q.z*o.y);
    float ty = 2.0f*(q.z*o.x - q.x*o.z);
    float tz = 2.0f*(q.x*o.y - q.y*o.x);
    return mk3(o.x + qw*tx + q.y*tz - q.z*ty,
               o.y + qw*ty + q.z*tx - q.x*tz,
               o.z + qw*tz + q.x*ty - q.y*tx);
}

// hitInst: record index of the instance the hit triangle was reached through, -1 = none (only read under FEAT_INSTANCES)
template<uint32_t M>
PT_DEV void intersectionInfo(const DeviceScene &s, const RayD &ray, float4 hit, Info &info, int hitInst = -1)
{
    int ri = __float_as_int(hit.w);
    const uint32_t rj = (uint32_t)ri;
    float4 r0 = ld4(s.recs, rj*3u + 0u), r1 = ld4(s.recs, rj*3u + 1u), r2 = ld4(s.recs, rj*3u + 2u);
    // the attribute gather does not wait for the record to say "triangle": both are issued together (tri_attrs has an entry
    // for every record), which takes one memory round trip out of the shading chain
    float4 a0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), a1 = a0, a2 = a0, a3 = a0;
    if (M & FEAT_TRIANGLES) { a0 = ld4(s.tri_attrs, rj*4u + 0u); a1 = ld4(s.tri_attrs, rj*4u + 1u); a2 = ld4(s.tri_attrs, rj*4u + 2u); a3 = ld4(s.tri_attrs, rj*4u + 3u); }
    uint32_t meta = __float_as_uint(r0.w);
    int objIdx = (int)TGHIP_REC_OBJECT(meta);
    const TgHipObject &o = s.objects[objIdx];
    info.object = objIdx;
    info.p = ray.o + ray.d*hit.x;                      /* TraceableScene.hpp:184 */
    info.T = info.B = splat3(0.0f);
    info.hasTB = false;
    uint32_t kind = TGHIP_REC_KIND(meta);
    if ((M & FEAT_TRIANGLES) && kind == TGHIP_REC_TRIANGLE) {   /* TriangleMesh.cpp:317-355, 80-106 */
        f3 NgU = cross(xyz(r1), xyz(r2));
        f3 dLocal = ray.d;
        if ((M & FEAT_INSTANCES) && hitInst >= 0) {    /* the master was intersected in its own space (Instance.cpp:295-297) */
            float4 q = ld4(s.recs, (uint32_t)hitInst*3u + 1u);
            dLocal = quatRotate(q.w, -xyz(q), ray.d);
        }
        info.backSide = dot(NgU, dLocal) > 0.0f;
        info.Ng = normalized(NgU);
        float u = hit.y, v = hit.z;
        if (o.flags & TGHIP_OBJF_SMOOTH) {
            f3 n0 = mk3(a0.x, a0.y, a0.z), n1 = mk3(a0.w, a1.x, a1.y), n2 = mk3(a1.z, a1.w, a2.x);
            info.Ns = normalized(n0*(1.0f - u - v) + n1*u + n2*v);
        } else {
            info.Ns = info.Ng;
        }
        info.u = (1.0f - u - v)*a2.y + u*a2.w + v*a3.y;
        info.v = (1.0f - u - v)*a2.z + u*a3.x + v*a3.z;
        info.bsdf = __float_as_int(a3.w);
        if constexpr ((M & FEAT_BUMP) != 0u) {         /* TriangleMesh::tangentSpace (:362-384); the record holds p0, p1 - p0, p2 - p0 */
            const f3 q1 = xyz(r1), q2 = xyz(r2);
            const float s1 = a2.w - a2.y, t1 = a3.x - a2.z;
            const float s2 = a3.y - a2.y, t2 = a3.z - a2.z;
            const float invDet = s1*t2 - s2*t1;
            if (!(fabsf(invDet) < 1e-6f) && !((M & FEAT_INSTANCES) && hitInst >= 0)) {
                info.T = normalized(q1*t2 - q2*t1);
                info.B = normalized(q2*s1 - q1*s2);
                info.hasTB = true;
            }
        }
    } else if (kind == TGHIP_REC_QUAD) {               /* Quad.cpp:123-131 */
        info.Ng = info.Ns = ld3(o.normal);
        info.u = hit.y; info.v = hit.z;
        info.bsdf = o.bsdf;
        info.backSide = dot(ray.d, info.Ng) >= 0.0f;
        if constexpr ((M & FEAT_BUMP) != 0u) { info.T = ld3(o.edge0); info.B = ld3(o.edge1); info.hasTB = true; }   /* Quad.cpp:133-139 */
    } else if ((M & FEAT_CYLINDER) && kind == TGHIP_REC_CYLINDER) {   /* Cylinder.cpp:122-132 */
        cylinderSurface(o, ray, hit.x, hit.z, info.Ng, info.u, info.v);
        info.Ns = info.Ng;
        info.bsdf = o.bsdf;
        info.backSide = hit.y != 0.0f;
        if constexpr ((M & FEAT_BUMP) != 0u) { info.T = ld3(o.normal); info.B = cross(info.Ng, info.T); info.hasTB = true; }   /* Cylinder.cpp:135-141: T = _axis */
    } else if ((M & FEAT_SOLIDS) && kind == TGHIP_REC_DISK) {    /* Disk.cpp:114-129 */
        info.Ng = info.Ns = ld3(o.normal);
        diskSurface(o, info.p, hit.z, info.u, info.v);
        info.bsdf = o.bsdf;
        info.backSide = hit.y != 0.0f;
        if constexpr ((M & FEAT_BUMP) != 0u) {         /* Disk::tangentSpace (:129-140) */
            f3 dd = info.p - ld3(o.pos);
            if (lengthSq(dd) != 0.0f) {
                dd = normalized(dd);
                info.T = cross(ld3(o.normal), dd);
                info.B = dd;
                info.hasTB = true;
            }
        }
    } else if (!(M & FEAT_SOLIDS) || kind == TGHIP_REC_CUBE) {   /* Cube.cpp:157-170 */
        cubeSurface(o, info.p, info.Ng, info.u, info.v);
        info.Ns = info.Ng;
        info.bsdf = o.bsdf;
        info.backSide = hit.y != 0.0f;
        if constexpr ((M & FEAT_BUMP) != 0u) {         /* Cube::tangentSpace (:172-182) */
            const f3 lp = mat3TMul(o.rot, info.p - ld3(o.pos));
            const float ex[3] = {fabsf(lp.x) - o.scale[0], fabsf(lp.y) - o.scale[1], fabsf(lp.z) - o.scale[2]};
            int dim = 0;
            if (ex[1] > ex[dim]) dim = 1;
            if (ex[2] > ex[dim]) dim = 2;
            const int dt = (dim + 1) % 3, db = (dim + 2) % 3;
            info.T = mat3Mul(o.rot, mk3(dt == 0 ? 1.0f : 0.0f, dt == 1 ? 1.0f : 0.0f, dt == 2 ? 1.0f : 0.0f));
            info.B = mat3Mul(o.rot, mk3(db == 0 ? 1.0f : 0.0f, db == 1 ? 1.0f : 0.0f, db == 2 ? 1.0f : 0.0f));
            info.hasTB = true;
        }
    } else {                                           /* Sphere.cpp:120-129 */
        sphereSurface(o, info.p, info.Ng, info.u, info.v);
        info.Ns = info.Ng;
        info.bsdf = o.bsdf;
        info.backSide = hit.y != 0.0f;
        if constexpr ((M & FEAT_BUMP) != 0u) {         /* Sphere::tangentSpace (:131-137) */
            const f3 localN = mat3TMul(o.rot, info.Ng);
            info.T = mat3Mul(o.rot, mk3(-localN.y, localN.x, localN.z));
            info.B = cross(info.Ns, info.T);
            info.hasTB = true;
        }
    }
    if ((M & FEAT_INSTANCES) && hitInst >= 0) {
        /* Instance::intersectionInfo (primitives/Instance.cpp:337-346): normals to world space; info.p -- already the
         * WORLD-space hit point (TraceableScene.hpp:184) -- is transformed once more, as the reference does */
        float4 i0 = ld4(s.recs, (uint32_t)hitInst*3u + 0u), i1 = ld4(s.recs, (uint32_t)hitInst*3u + 1u);
        info.Ng = quatRotate(i1.w, xyz(i1), info.Ng);
        info.Ns = quatRotate(i1.w, xyz(i1), info.Ns);
        info.p = xyz(i0) + quatRotate(i1.w, xyz(i1), info.p);
        info.object = (int)TGHIP_REC_OBJECT(__float_as_uint(i0.w));
        info.hasTB = false;                            /* Instance::tangentSpace (Instance.cpp:348-351) */
    }
}

// BitmapTexture::derivatives (textures/BitmapTexture.cpp:359-398) of a scalar bitmap (bump maps are requested as scalars, Bsdf.cpp:24):
// central differences of the four texels around the lookup, interpolated; constant and checker textures have none
PT_DEV void textureDerivatives(const DeviceScene &s, int texIdx, float u0, float v0, float &du, float &dv)
{
    const TgHipTexture &t = s.textures[texIdx];
    du = dv = 0.0f;
    if (t.type != TGHIP_TEX_BITMAP)
        return;
    const int w = t.w, h = t.h;
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
    auto texel = [&](int x, int y) { const f3 c = bitmapTexel(s, t, x, y); return (t.flags & TGHIP_TEXF_RGB) ? (c.x + c.y + c.z)/3.0f : c.x; };
    const float a01 = texel(x1, y0), a02 = texel(x2, y0);
    const float a10 = texel(x0, y1), a11 = texel(x1, y1), a12 = texel(x2, y1), a13 = texel(x3, y1);
    const float a20 = texel(x0, y2), a21 = texel(x1, y2), a22 = texel(x2, y2), a23 = texel(x3, y2);
    const float a31 = texel(x1, y3), a32 = texel(x2, y3);
    const float du11 = a12 - a10, du12 = a13 - a11, du21 = a22 - a20, du22 = a23 - a21;
    const float dv11 = a21 - a01, dv21 = a31 - a11, dv12 = a22 - a02, dv22 = a32 - a12;
    du = ((du11*(1.0f - u) + du12*u)*(1.0f - v) + (du21*(1.0f - u) + du22*u)*v)*t.scale;
    dv = ((dv11*(1.0f - u) + dv12*u)*(1.0f - v) + (dv21*(1.0f - u) + dv22*u)*v)*t.scale;
}

// Primitive::setupTangentFrame (primitives/Primitive.cpp:125-163): the frame of the shading normal -- unless the bsdf carries a non-constant
// bump map (TgHipBsdf::bump1; FEAT_BUMP variants): then tangent and bitangent come from the primitive's tangent space, tilted by the map's
// derivatives.  (Anisotropic lobes, the other reason for the long way, belong to the hair bcsdfs.)
template<uint32_t M>
PT_DEV Frame shadingFrame(const DeviceScene &s, const Info &info)
{
    if constexpr ((M & FEAT_BUMP) != 0u) {
        const int bump = s.bsdfs[info.bsdf].bump1 - 1;
        if (bump >= 0 && info.hasTB) {
            f3 T = info.T, B = info.B, N = info.Ns;
            float du, dv;
            textureDerivatives(s, bump, info.u, info.v, du, dv);
            T = T + info.Ns*(du - dot(info.Ns, T));
            B = B + info.Ns*(dv - dot(info.Ns, B));
            N = cross(T, B);
            if (!(N.x == 0.0f && N.y == 0.0f && N.z == 0.0f)) {
                if (dot(N, info.Ns) < 0.0f)
                    N = -N;
                N = normalized(N);
                T = T - N*dot(N, T);
                if (!(T.x == 0.0f && T.y == 0.0f && T.z == 0.0f)) {
                    Frame f;
                    f.normal = N; f.tangent = normalized(T); f.bitangent = cross(N, f.tangent);
                    return f;
                }
            }
        }
    }
    return frameFromNormal(info.Ns);
}

// ---------------------------------------------------------------------------------------------
// Lights (Quad.cpp:172-187,216-223,235-238,256-279; InfiniteSphere.cpp:27-51,161-176,218-229,241-244,261-266)
// ---------------------------------------------------------------------------------------------
PT_DEV void infDirectionToUV(const TgHipObject &o, f3 wi, float &u, float &v, float &sinTheta)
{
    // (a skydome maps world directions to its image as they are, Skydome.cpp:41-50; not through an identity matrix: -0 + 0 = +0 would move
    // atan2f's branch cut)
    f3 wLocal = (o.flags & TGHIP_OBJF_SKYDOME) ? wi : mat3TMul(o.rot, wi);
    sinTheta = sqrtf(fmaxf(1.0f - wLocal.y*wLocal.y, 0.0f));
    u = atan2fH(wLocal.z, wLocal.x)*PT_INV_TWO_PI + 0.5f;
    v = acosfExact(-wLocal.y)*PT_INV_PI;
}
PT_DEV f3 infUvToDirection(const TgHipObject &o, float u, float v, float &sinTheta)
{
    float phi = (u - 0.5f)*PT_TWO_PI;
    float theta = v*PT_PI;
    float cosTheta, sinPhi, cosPhi;
    sincosfH(theta, sinTheta, cosTheta);
    sincosfH(phi, sinPhi, cosPhi);
    const f3 wLocal = mk3(cosPhi*sinTheta, -cosTheta, sinPhi*sinTheta);
    return (o.flags & TGHIP_OBJF_SKYDOME) ? wLocal : mat3Mul(o.rot, wLocal);     // (Skydome.cpp:51-61)
}

struct LightHit { float t, u, v; bool backSide; f3 n; float sinTheta; };   /* n: surface normal at the hit (cube lights); sinTheta: infinite sphere (below) */

/* light.intersect(ray) + intersectionInfo: analytic hit test that precedes the shadow ray (TraceBase.cpp:155-162) */
template<uint32_t M>
PT_DEV bool lightIntersect(const DeviceScene &s, int objIdx, const RayD &ray, LightHit &lh)
{
    const TgHipObject &o = s.objects[objIdx];
    lh.n = splat3(0.0f);
    if (!(M & (FEAT_INFINITE | FEAT_SOLIDS)) || o.type == TGHIP_OBJ_QUAD) {
        f3 n = ld3(o.normal);
        if (!quadTest(ld3(o.base), ld3(o.edge0), ld3(o.edge1), o.inv_uv_sq[0], o.inv_uv_sq[1], n, ray, ray.tmax, lh.t, lh.u, lh.v))
            return false;
        lh.backSide = dot(ray.d, n) >= 0.0f;
        return true;
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_CUBE) {       /* Cube::intersect + intersectionInfo */
        if (!cubeTest(&o, ray, ray.tmax, lh.t, lh.backSide)) return false;
        cubeSurface(o, ray.o + ray.d*lh.t, lh.n, lh.u, lh.v);
        return true;
    }
    if ((M & FEAT_CYLINDER) && o.type == TGHIP_OBJ_CYLINDER) {   /* Cylinder::intersect + intersectionInfo */
        float cap;
        if (!cylinderTest(&o, ray, ray.tmax, lh.t, lh.backSide, cap)) return false;
        cylinderSurface(o, ray, lh.t, cap, lh.n, lh.u, lh.v);
        return true;
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_DISK) {       /* Disk::intersect + intersectionInfo */
        float rSq;
        if (!diskTest(&o, ray, ray.tmax, lh.t, rSq, lh.backSide)) return false;
        diskSurface(o, ray.o + ray.d*lh.t, rSq, lh.u, lh.v);
        lh.n = ld3(o.normal);
        return true;
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_SPHERE) {     /* Sphere::intersect + intersectionInfo */
        if (!sphereTest(&o, ray, ray.tmax, lh.t, lh.backSide)) return false;
        sphereSurface(o, ray.o + ray.d*lh.t, lh.n, lh.u, lh.v);
        return true;
    }
    if (o.type == TGHIP_OBJ_INFINITE_SPHERE_CAP) {            /* InfiniteSphereCap::intersect + intersectionInfo (:60-90) */
        if (dot(ray.d, ld3(o.normal)) < o.scale[0]) return false;
        lh.t = ray.tmax; lh.backSide = false; lh.u = 0.0f; lh.v = 0.0f;
        return true;
    }
    lh.t = ray.tmax; lh.backSide = false;
    infDirectionToUV(o, ray.d, lh.u, lh.v, lh.sinTheta);
    return true;
}
template<uint32_t M>
PT_DEV f3 lightEvalDirect(const DeviceScene &s, int objIdx, float u, float v, bool backSide)
{
    const TgHipObject &o = s.objects[objIdx];
    if (o.emission < 0 || backSide) return splat3(0.0f);
    return textureEval<M>(s, o.emission, u, v);
}
/* directPdf of the light for direction w from p; lh = lightIntersect's result for that ray */
template<uint32_t M>
PT_DEV float lightDirectPdf(const DeviceScene &s, int objIdx, f3 w, f3 p, const LightHit &lh)
{
    const TgHipObject &o = s.objects[objIdx];
    if (!(M & (FEAT_INFINITE | FEAT_SOLIDS)) || o.type == TGHIP_OBJ_QUAD) {
        f3 n = ld3(o.normal);
        float cosTheta = fabsf(dot(n, w));
        float t = dot(n, ld3(o.base) - p)/dot(n, w);
        return t*t/(cosTheta*o.area);
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_CUBE) {       /* Cube.cpp:291-295 */
        f3 hp = p + w*lh.t;
        return lengthSq(p - hp)/(-dot(w, lh.n)*o.area);
    }
    if ((M & FEAT_CYLINDER) && o.type == TGHIP_OBJ_CYLINDER) {   /* Cylinder::directPdf (Cylinder.cpp:246-250) */
        f3 hp = p + w*lh.t;
        return lengthSq(p - hp)/(-dot(w, lh.n)*o.area);
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_DISK) {       /* Disk::directPdf (Disk.cpp:228-235) */
        f3 n = ld3(o.normal);
        float cosTheta = fabsf(dot(n, w));
        float t = dot(n, ld3(o.pos) - p)/dot(n, w);
        return t*t/(cosTheta*o.scale[0]*o.scale[0]*PT_PI);
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_SPHERE) {     /* Sphere.cpp:216-222 */
        float dist = length(ld3(o.pos) - p);
        float cosTheta = sqrtf(fmaxf(dist*dist - o.scale[0]*o.scale[0], 0.0f))/dist;
        return PT_INV_TWO_PI/(1.0f - cosTheta);
    }
    if (o.type == TGHIP_OBJ_INFINITE_SPHERE_CAP)               /* uniformSphericalCapPdf (InfiniteSphereCap.cpp:214-218) */
        return PT_INV_TWO_PI/(1.0f - o.scale[0]);
    if (o.type == TGHIP_OBJ_POINT)                             /* Point::directPdf (Point.cpp:117-121) */
        return lengthSq(p - ld3(o.pos));
    const TgHipTexture &t = s.textures[o.emission];
    if (!(M & FEAT_BITMAP) || t.type != TGHIP_TEX_BITMAP)
        return PT_INV_FOUR_PI;
    // (u, v, sinTheta) = infDirectionToUV(o, w): lightIntersect computed exactly that for this direction (InfiniteSphere::directPdf,
    // InfiniteSphere.cpp:218-229, maps the direction again; same function of the same argument, same result)
    return PT_INV_PI*PT_INV_TWO_PI*bitmapPdf(s, o.emission, t, lh.u, lh.v)/lh.sinTheta;
}
template<uint32_t M>
PT_DEV bool lightSampleDirect(const DeviceScene &s, int objIdx, f3 p, Rng &rng, f3 &d, float &dist, float &pdf)
{
    const TgHipObject &o = s.objects[objIdx];
    if ((M & FEAT_MESHLIGHT) && o.type == TGHIP_OBJ_MESH) {    /* TriangleMesh::sampleDirect / samplePosition (TriangleMesh.cpp:411-462) */
        const float *cdf = s.light_tris + o.first_light_tri;
        const float *tris = cdf + o.num_light_tris + 1;
        float u = RNG1D(rng);
        int idx = upperBoundIdx(cdf, o.num_light_tris + 1, u) - 1;     /* Distribution1D::warp */
        const float *t = tris + (size_t)idx*9;
        f3 p0 = ld3(t), p1 = ld3(t + 3), p2 = ld3(t + 6);
        f3 normal = normalized(cross(p1 - p0, p2 - p0));
        float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
        float uSqrt = sqrtf(xi0);                                      /* SampleWarp::uniformTriangleUv */
        float alpha = 1.0f - uSqrt, beta = (1.0f - xi1)*uSqrt;
        f3 q = p0*alpha + p1*beta + p2*(1.0f - alpha - beta);
        f3 L = q - p;
        float rSq = lengthSq(L);
        dist = sqrtf(rSq);
        d = L/dist;
        float cosTheta = -dot(normal, d);
        if (cosTheta <= 0.0f)
            return false;
        pdf = rSq/(cosTheta*o.area);
        return true;
    }
    if ((M & FEAT_CYLINDER) && o.type == TGHIP_OBJ_CYLINDER) {   /* Cylinder::sampleDirect + samplePosition (Cylinder.cpp:149-170, 181-196) */
        const float radius = o.scale[0], halfHeight = o.scale[1];
        f3 ng, q;
        if (o.scale[2] != 0.0f && rngNextBoolean(rng, PT_TWO_PI*sqr(radius)*o.inv_area)) {
            float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
            float phi = xi0*PT_TWO_PI, rr = sqrtf(xi1);                    /* SampleWarp::uniformDisk */
            float sign = rngNextBoolean(rng, 0.5f) ? -1.0f : 1.0f;
            ng = mk3(0.0f, sign, 0.0f);
            float sinPhi, cosPhi;
            sincosfH(phi, sinPhi, cosPhi);
            q = mk3(cosPhi*rr*radius, sign*halfHeight, sinPhi*rr*radius);
        } else {
            float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
            float phi = xi0*PT_TWO_PI;                                     /* SampleWarp::uniformCylinder */
            float cx, cy, cz = xi1*2.0f - 1.0f;
            sincosfH(phi, cy, cx);
            ng = mk3(cx, 0.0f, cy);
            q = mk3(cx*radius, cz*halfHeight, cy*radius);
        }
        ng = mat3Mul(o.rot, ng);
        q = mat3Mul(o.rot, q) + ld3(o.pos);
        f3 L = q - p;
        float rSq = lengthSq(L);
        dist = sqrtf(rSq);
        d = L/dist;
        float cosTheta = -dot(ng, d);
        if (cosTheta <= 0.0f)
            return false;
        pdf = rSq/(cosTheta*o.area);
        return true;
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_DISK) {       /* Disk::sampleDirect (Disk.cpp:178-194) */
        f3 n = ld3(o.normal), center = ld3(o.pos);
        if (dot(n, p - center) < 0.0f)
            return false;
        float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
        float phi = xi0*PT_TWO_PI, rr = sqrtf(xi1);                    /* SampleWarp::uniformDisk */
        float sinPhi, cosPhi;
        sincosfH(phi, sinPhi, cosPhi);
        float lx = cosPhi*rr*o.scale[0], ly = sinPhi*rr*o.scale[0];
        f3 q = center + ld3(o.edge1)*lx + ld3(o.edge0)*ly;
        f3 L = q - p;
        float rSq = lengthSq(L);
        dist = sqrtf(rSq);
        d = L/dist;
        if (-dot(d, n) < o.scale[1])
            return false;
        float cosTheta = -dot(n, d);
        pdf = rSq/(cosTheta*o.scale[0]*o.scale[0]*PT_PI);
        return true;
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_CUBE) {       /* Cube::sampleDirect / samplePosition / sampleFace (Cube.cpp:229-245,189-213,42-55) */
        float u = RNG1D(rng);
        int dim;
        u *= o.face_cdf[2];
        if (u < o.face_cdf[0]) { u /= o.face_cdf[0]; dim = 0; }
        else if (u < o.face_cdf[1]) { u = (u - o.face_cdf[0])/(o.face_cdf[1] - o.face_cdf[0]); dim = 1; }
        else { u = (u - o.face_cdf[1])/(o.face_cdf[2] - o.face_cdf[1]); dim = 2; }
        float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
        float sgn = u < 0.5f ? -1.0f : 1.0f;
        float a = (xi0*2.0f - 1.0f), b = (xi1*2.0f - 1.0f);
        // p[dim] = sgn*scale[dim], p[(dim+1)%3] = a*scale[..], p[(dim+2)%3] = b*scale[..]
        f3 pp = dim == 0 ? mk3(sgn*o.scale[0], a*o.scale[1], b*o.scale[2])
              : dim == 1 ? mk3(b*o.scale[0], sgn*o.scale[1], a*o.scale[2])
                         : mk3(a*o.scale[0], b*o.scale[1], sgn*o.scale[2]);
        f3 nn = dim == 0 ? mk3(sgn, 0.0f, 0.0f) : dim == 1 ? mk3(0.0f, sgn, 0.0f) : mk3(0.0f, 0.0f, sgn);
        f3 q = mat3Mul(o.rot, pp) + ld3(o.pos);
        f3 Ng = mat3Mul(o.rot, nn);
        f3 L = q - p;
        float rSq = lengthSq(L);
        dist = sqrtf(rSq);
        d = L/dist;
        float cosTheta = -dot(Ng, d);
        if (cosTheta <= 0.0f)
            return false;
        pdf = rSq/(cosTheta*o.area);
        return true;
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_SPHERE) {     /* Sphere::sampleDirect (Sphere.cpp:173-194) */
        f3 L = ld3(o.pos) - p;
        float dd = length(L);
        float C = dd*dd - o.scale[0]*o.scale[0];
        if (C <= 0.0f)
            return false;
        L = normalized(L);
        float cosTheta = sqrtf(C)/dd;
        float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
        float phi = xi0*PT_TWO_PI;                             /* SampleWarp::uniformSphericalCap */
        float z = xi1*(1.0f - cosTheta) + cosTheta;
        float r = sqrtf(fmaxf(1.0f - z*z, 0.0f));
        float sinPhi, cosPhi;
        sincosfH(phi, sinPhi, cosPhi);
        f3 local = mk3(cosPhi*r, sinPhi*r, z);
        float B = dd*local.z;
        float det = sqrtf(fmaxf(B*B - C, 0.0f));
        dist = B - det;
        d = toGlobal(frameFromNormal(L), local);
        pdf = PT_INV_TWO_PI/(1.0f - cosTheta);
        return true;
    }
    if (!(M & FEAT_INFINITE) || o.type == TGHIP_OBJ_QUAD) {
        f3 n = ld3(o.normal);
        if (dot(n, p - ld3(o.base)) <= 0.0f)
            return false;
        float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
        f3 q = ld3(o.base) + ld3(o.edge0)*xi0 + ld3(o.edge1)*xi1;
        f3 dd = q - p;
        float rSq = lengthSq(dd);
        dist = sqrtf(rSq);
        dd = dd/dist;
        float cosTheta = -dot(n, dd);
        pdf = rSq/(cosTheta*o.area);
        d = dd;
        return true;
    }
    if (o.type == TGHIP_OBJ_POINT) {                          /* Point::sampleDirect (Point.cpp:93-101): draws nothing */
        f3 L = ld3(o.pos) - p;
        float rSq = lengthSq(L);
        dist = sqrtf(rSq);
        d = L/dist;
        pdf = rSq;
        return true;
    }
    if (o.type == TGHIP_OBJ_INFINITE_SPHERE_CAP) {            /* InfiniteSphereCap::sampleDirect (:130-138) */
        float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
        float phi = xi0*PT_TWO_PI;                                     /* SampleWarp::uniformSphericalCap */
        float z = xi1*(1.0f - o.scale[0]) + o.scale[0];
        float r = sqrtf(fmaxf(1.0f - z*z, 0.0f));
        float sinPhi, cosPhi;
        sincosfH(phi, sinPhi, cosPhi);
        f3 local = mk3(cosPhi*r, sinPhi*r, z);
        d = ld3(o.edge0)*local.x + ld3(o.edge1)*local.y + ld3(o.normal)*local.z;   /* TangentFrame::toGlobal */
        dist = PT_INF;
        pdf = PT_INV_TWO_PI/(1.0f - o.scale[0]);
        return true;
    }
    const TgHipTexture &t = s.textures[o.emission];
    float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
    dist = PT_INF;
    if (!(M & FEAT_BITMAP) || t.type != TGHIP_TEX_BITMAP) {
        d = uniformSphere(xi0, xi1);
        pdf = PT_INV_FOUR_PI;
        return true;
    }
    float u, v, sinTheta;
    BitmapPick pick;
    bitmapSample(s, o.emission, t, xi0, xi1, u, v, pick);
    d = infUvToDirection(o, u, v, sinTheta);
    pdf = PT_INV_PI*PT_INV_TWO_PI*bitmapPdf(s, o.emission, t, u, v, &pick)/sinTheta;
    return pdf != 0.0f;
}
template<uint32_t M>
PT_DEV float lightApproximateRadiance(const DeviceScene &s, int objIdx, f3 p)
{
    const TgHipObject &o = s.objects[objIdx];
    if ((M & FEAT_MESHLIGHT) && o.type == TGHIP_OBJ_MESH)      /* TriangleMesh.cpp:514-517: "unknown" */
        return -1.0f;
    if ((M & FEAT_CYLINDER) && o.type == TGHIP_OBJ_CYLINDER)   /* Cylinder.cpp:280-284: "unknown" too (the FEAT_CYLINDER variant has FEAT_MESHLIGHT's chooseLight) */
        return -1.0f;
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_CUBE) {       /* Cube.cpp:326-330 */
        f3 lp = mat3TMul(o.rot, p - ld3(o.pos));
        f3 ap = mk3(fmaxf(fabsf(lp.x), 0.0f), fmaxf(fabsf(lp.y), 0.0f), fmaxf(fabsf(lp.z), 0.0f));
        return max3(ld3(s.textures[o.emission].avg))*o.face_cdf[2]/lengthSq(ap);
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_DISK) {       /* Disk::approximateRadiance (Disk.cpp:253-281) */
        if (o.emission < 0) return 0.0f;
        f3 n = ld3(o.normal);
        f3 coneD = p - ld3(o.base);
        if (dot(coneD, n)/length(coneD) < o.scale[1])
            return 0.0f;
        f3 dd = ld3(o.pos) - p;
        f3 e0 = ld3(o.edge0)*o.scale[0], e1 = ld3(o.edge1)*o.scale[0];
        f3 R0 = dd - e0 - e1;
        f3 R1 = R0 + e0*2.0f;
        f3 R2 = R1 + e1*2.0f;
        f3 R3 = R0 + e1*2.0f;
        f3 n0 = normalized(cross(R0, R1)), n1 = normalized(cross(R1, R2));
        f3 n2 = normalized(cross(R2, R3)), n3 = normalized(cross(R3, R0));
        float Q = acosfExact(dot(n0, n1)) + acosfExact(dot(n1, n2)) + acosfExact(dot(n2, n3)) + acosfExact(dot(n3, n0));
        return (PT_TWO_PI - fabsf(Q))*max3(ld3(s.textures[o.emission].avg));
    }
    if ((M & FEAT_SOLIDS) && o.type == TGHIP_OBJ_SPHERE) {     /* Sphere.cpp:266-271, 33-40 */
        if (o.emission < 0) return 0.0f;
        float dd = length(ld3(o.pos) - p);
        float cosTheta = sqrtf(fmaxf(dd*dd - o.scale[0]*o.scale[0], 0.0f))/dd;
        return PT_TWO_PI*(1.0f - cosTheta)*max3(ld3(s.textures[o.emission].avg));
    }
    if (!(M & FEAT_INFINITE) || o.type == TGHIP_OBJ_QUAD) {
        if (o.emission < 0) return 0.0f;
        f3 R0 = ld3(o.base) - p;
        if (dot(R0, ld3(o.normal)) >= 0.0f)
            return 0.0f;
        f3 R1 = R0 + ld3(o.edge0);
        f3 R2 = R1 + ld3(o.edge1);
        f3 R3 = R0 + ld3(o.edge1);
        f3 n0 = normalized(cross(R0, R1)), n1 = normalized(cross(R1, R2));
        f3 n2 = normalized(cross(R2, R3)), n3 = normalized(cross(R3, R0));
        float Q = acosfExact(dot(n0, n1)) + acosfExact(dot(n1, n2)) + acosfExact(dot(n2, n3)) + acosfExact(dot(n3, n0));
        return (PT_TWO_PI - fabsf(Q))*max3(ld3(s.textures[o.emission].avg));
    }
    if (o.type == TGHIP_OBJ_POINT) {                           /* Point::approximateRadiance (Point.cpp:166-169) */
        /* scale = Point::_power as prepareForRender left it (Point.cpp:186): 0 for a light given by "power" */
        return PT_INV_FOUR_PI*max3(ld3(o.scale))/lengthSq(ld3(o.pos) - p);
    }
    if (o.emission < 0 || !(o.flags & TGHIP_OBJF_SAMPLE)) return 0.0f;
    if (o.type == TGHIP_OBJ_INFINITE_SPHERE_CAP)               /* InfiniteSphereCap.cpp:220-225 */
        return PT_TWO_PI*(1.0f - o.scale[0])*max3(ld3(s.textures[o.emission].avg));
    if (o.flags & TGHIP_OBJF_SKYDOME)                          /* Skydome::approximateRadiance (Skydome.cpp:240-243) */
        return PT_FOUR_PI*max3(ld3(s.textures[o.emission].avg));
    return PT_TWO_PI*max3(ld3(s.textures[o.emission].avg));
}

/* TraceBase::chooseLight (TraceBase.cpp:416-459); returns the object index of the light or -1 */
template<uint32_t M>
PT_DEV int chooseLight(const DeviceScene &s, Rng &rng, f3 p, float &weight)
{
    int n = (int)s.num_lights;
    if (n == 0) return -1;
    if (!(M & FEAT_MULTILIGHT) || n == 1) { weight = 1.0f; return s.lights[0]; }
    n = min(n, 16);
    // Lights that answer "unknown" (a negative weight) receive the mean of the known weights (TraceBase.cpp:434-446): mesh and cylinder
    // emitters always do (-1), and a quad does whenever its solid angle -- 2 pi minus four arc cosines, Quad.cpp:253-281 -- rounds to
    // less than zero, which the few-mm emitters of the shipped non-exponential scene reach in about one light sample in 10^5.  Every
    // variant therefore runs the full rule.  The reference fills a pdf array in order; here the weights are re-evaluated instead of
    // stored per lane (a second time only when some light answered "unknown").
    float total = 0.0f;
    int numNonNegative = 0;
    for (int i = 0; i < n; ++i) {
        float w = lightApproximateRadiance<M>(s, s.lights[i], p);
        if (w >= 0.0f) { total += w; numNonNegative++; }
    }
    const bool allUnknown = numNonNegative == 0;
    float knownTotal = total;
    if (allUnknown) {
        total = (float)n;
    } else if (numNonNegative < n) {
        for (int i = 0; i < n; ++i)
            if (lightApproximateRadiance<M>(s, s.lights[i], p) < 0.0f)
                total += (total == 0.0f ? 1.0f : total)/numNonNegative;   // uses the running total, like the reference's loop
    }
    if (total == 0.0f) return -1;
    float t = RNG1D(rng)*total;
    float running = knownTotal;
    for (int i = 0; i < n; ++i) {
        float w = lightApproximateRadiance<M>(s, s.lights[i], p);
        float pdf;
        if (allUnknown) pdf = 1.0f;
        else if (w < 0.0f) { pdf = (running == 0.0f ? 1.0f : running)/numNonNegative; running += pdf; }
        else pdf = w;
        if (t < pdf || i == n - 1) { weight = total/pdf; return s.lights[i]; }
        t -= pdf;
    }
    return -1;
}

// ---- participating media: HomogeneousMedium (media/HomogeneousMedium.cpp:43-131) with the transmittances of
// transmittances/*.cpp.  Of MediumState only `bounce` is carried (path flags, pt_kernels.h): firstScatter == (bounce == 0). ----
PT_DEV int selectMedium(const TgHipObject &o, int current, bool geometricBackside)   /* Primitive.hpp:177-183 */
{
    if (o.int_medium >= 0 || o.ext_medium >= 0)
        return geometricBackside ? o.int_medium : o.ext_medium;
    return current;
}

/* The four kernels of a transmittance for one channel -- k: 0 = surfaceSurface, 1 = surfaceMedium, 2 = mediumSurface,
 * 3 = mediumMedium (transmittances/{Exponential,Linear,Quadratic,DoubleExponential,Pulse,Erlang}Transmittance.cpp) */
PT_DEV float transLeafKernel(const TgHipMedium &m, int k, float tau)
{
#ifdef PT_EXP_TRANS_ONLY     /* compile-only / A-B diagnostic: what the media kernels cost without the eight non-exponential transmittances */
    return fmathExp(-tau);
#endif
    const float p0 = m.trans_p[0], p1 = m.trans_p[1], p2 = m.trans_p[2];
    switch (m.trans_type) {
    case TGHIP_TRANS_LINEAR: {                          /* LinearTransmittance.cpp:32-57 */
        if (k == 0) return 1.0f - fminf(tau/p0, 1.0f);
        if (k == 1) return tau > p0 ? 0.0f : 1.0f/p0;
        if (k == 2) return tau > p0 ? 0.0f : 1.0f;
        return fabsf(tau - p0) < 1e-3f ? 1.0f : 0.0f;
    }
    case TGHIP_TRANS_QUADRATIC: {                       /* QuadraticTransmittance.cpp:32-52 */
        float t = fminf(tau/p0, 1.0f);
        if (k == 0) return 1.0f - 2.0f*t + t*t;
        if (k == 1) return (2.0f/p0)*(1.0f - t);
        if (k == 2) return 1.0f - t;
        return tau > p0 ? 0.0f : 1.0f/p0;
    }
    case TGHIP_TRANS_DOUBLE_EXPONENTIAL: {              /* DoubleExponentialTransmittance.cpp:34-49 */
        float ea = expfH(-p0*tau), eb = expfH(-p1*tau);
        if (k == 0) return 0.5f*(ea + eb);
        if (k == 1) return 0.5f*(p0*ea + p1*eb);
        if (k == 2) return (p0*ea + p1*eb)/(p0 + p1);
        return (sqr(p0)*ea + sqr(p1)*eb)/(p0 + p1);
    }
    case TGHIP_TRANS_PULSE: {                           /* PulseTransmittance.cpp:45-82 */
        const float a = p0, b = p1, n = p2;
        const int num = (int)n;
        if (k == 0) {
            float idxF = fminf(fmaxf(n*(tau - a)/(b - a) + 0.5f, 0.0f), n);
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
        float e = expfH(-p0*tau);
        if (k == 0) return 0.5f*e*(2.0f + p0*tau);
        if (k == 1) return e*(1.0f + p0*tau)*p0*0.5f;
        if (k == 2) return e*(1.0f + p0*tau);
        return sqr(p0)*tau*e;
    }
    case TGHIP_TRANS_DAVIS: {                           /* DavisTransmittance.cpp:34-49 */
        if (k == 0) return powfH(1.0f + tau/p0, -p0);
        if (k == 1 || k == 2) return powfH(1.0f + tau/p0, -(p0 + 1.0f));
        return (1.0f + 1.0f/p0)*powfH(1.0f + tau/p0, -(p0 + 2.0f));
    }
    case TGHIP_TRANS_DAVIS_WEINSTEIN: {                 /* DavisWeinsteinTransmittance.cpp:39-82; NaN -> 0 */
        float beta = 2.0f*p0 - 1.0f;
        float alpha = powfH(tau, 1 - beta)/powfH(p1, 1 + beta);
        float base = 1.0f + tau/alpha;
        float trSurface = powfH(base, -alpha), Tr;
        if (k == 0) {
            Tr = trSurface;
        } else if (k == 1 || k == 2) {
            Tr = trSurface*(beta/base - (beta - 1.0f)*alpha/tau*logfH(base));
        } else {
            float logBase = logfH(base);
            float term1 = beta*(-1.0f + beta*(1.0f + tau) + (-1.0f + 2.0f*beta)*tau/alpha)/(tau*base*base);
            float term2 = ((-1.0f + beta)*beta*alpha/(tau*tau)*(2.0f*tau + base)*logBase)/base;
            float term3 = (beta - 1.0f)*alpha/tau*logBase;
            Tr = trSurface*(term1 - term2 + term3*term3);
        }
        return isnan(Tr) ? 0.0f : Tr;
    }
    default:                                            /* ExponentialTransmittance.cpp:26-41: FastMath::exp */
        return fmathExp(-tau);
    }
}
PT_DEV float transLeafSigmaBar(const TgHipMedium &m)
{
#ifdef PT_EXP_TRANS_ONLY
    return 1.0f;
#endif
    switch (m.trans_type) {
    case TGHIP_TRANS_LINEAR: return 1.0f/m.trans_p[0];
    case TGHIP_TRANS_QUADRATIC: return 2.0f/m.trans_p[0];
    case TGHIP_TRANS_DOUBLE_EXPONENTIAL: return 0.5f*(m.trans_p[0] + m.trans_p[1]);
    case TGHIP_TRANS_PULSE: return 2.0f/(m.trans_p[1] - m.trans_p[0]);
    case TGHIP_TRANS_ERLANG: return m.trans_p[0]*0.5f;
    default: return 1.0f;
    }
}
PT_DEV f3 transLeafKernel3(const TgHipMedium &m, int k, f3 tau)
{
    if (m.trans_type == TGHIP_TRANS_DAVIS_WEINSTEIN)    /* evaluated on the first channel only and broadcast (:46-49) */
        return splat3(transLeafKernel(m, k, tau.x));
    return mk3(transLeafKernel(m, k, tau.x), transLeafKernel(m, k, tau.y), transLeafKernel(m, k, tau.z));
}
/* InterpolatedTransmittance (InterpolatedTransmittance.cpp:34-72): its operands are the two media[] entries behind it */
PT_DEV float lerpf(float a, float b, float u) { return a*(1.0f - u) + b*u; }
PT_DEV bool transIsDirac(const TgHipMedium &m) { return m.trans_type == TGHIP_TRANS_LINEAR || m.trans_type == TGHIP_TRANS_PULSE; }
PT_DEV float transSigmaBar(const TgHipMedium &m)
{
    if (m.trans_type != TGHIP_TRANS_INTERPOLATED) return transLeafSigmaBar(m);
    return 1.0f/lerpf(1.0f/transLeafSigmaBar((&m)[1]), 1.0f/transLeafSigmaBar((&m)[2]), m.trans_p[0]);
}
PT_DEV float transInterpolate(const TgHipMedium &m, int k, float a, float b)   /* a, b: the operands' kernel k (mediumSurface for k = 1) */
{
    const float u = m.trans_p[0];
    if (k == 0) return transSigmaBar(m)*lerpf(a/transLeafSigmaBar((&m)[1]), b/transLeafSigmaBar((&m)[2]), u);
    if (k == 1) return lerpf(a, b, u)*transSigmaBar(m);
    if (k == 2) return lerpf(a, b, u);
    bool diracA = transIsDirac((&m)[1]) && a > 0.0f, diracB = transIsDirac((&m)[2]) && b > 0.0f;
    if (diracA != diracB) return diracA ? a : b;
    return lerpf(a, b, u);
}
PT_DEV f3 transKernel3(const TgHipMedium &m, int k, f3 tau)
{
    if (m.trans_type != TGHIP_TRANS_INTERPOLATED) return transLeafKernel3(m, k, tau);
    const int kk = k == 1 ? 2 : k;
    f3 a = transLeafKernel3((&m)[1], kk, tau), b = transLeafKernel3((&m)[2], kk, tau);
    return mk3(transInterpolate(m, k, a.x, b.x), transInterpolate(m, k, a.y, b.y), transInterpolate(m, k, a.z, b.z));
}
PT_DEV f3 transEval(const TgHipMedium &m, f3 tau, bool startOnSurface, bool endOnSurface)   /* Transmittance::eval (Transmittance.hpp:22-30) */
{
    if (startOnSurface && endOnSurface) return transKernel3(m, 0, tau);
    if (!startOnSurface && !endOnSurface) return transKernel3(m, 3, tau)/transSigmaBar(m);
    return transKernel3(m, 2, tau);
}
template<uint32_t M>
PT_DEV float transLeafSample(const TgHipMedium &m, Rng &rng, bool startOnSurface)   /* sampleSurface / sampleMedium */
{
#ifdef PT_EXP_TRANS_ONLY
    return -logfH(1.0f - RNG1D(rng));
#endif
    const float p0 = m.trans_p[0], p1 = m.trans_p[1], p2 = m.trans_p[2];
    switch (m.trans_type) {
    case TGHIP_TRANS_LINEAR:
        return startOnSurface ? p0*RNG1D(rng) : p0;
    case TGHIP_TRANS_QUADRATIC:
        return startOnSurface ? p0*(1.0f - sqrtf(1.0f - RNG1D(rng))) : p0*RNG1D(rng);
    case TGHIP_TRANS_DOUBLE_EXPONENTIAL: {
        float t = -logfH(1.0f - RNG1D(rng));
        return rngNextBoolean(rng, startOnSurface ? 0.5f : p0/(p0 + p1)) ? t/p0 : t/p1;
    }
    case TGHIP_TRANS_PULSE: {
        const float a = p0, b = p1, n = p2;
        const int num = (int)n;
        if (!startOnSurface)
            return a + (0.5f + (float)(int)(RNG1D(rng)*n))/n*(b - a);
        float xi = RNG1D(rng)*n*0.5f;
        float delta = 1.0f/n;
        for (int i = 0; i < num; ++i) {
            float h0 = 1.0f - ((float)i + 0.0f)*delta;
            float h1 = 1.0f - ((float)i + 1.0f)*delta;
            xi -= h0*0.5f;
            if (xi < 0.0f)
                return a + ((float)i + 0.0f + 0.5f*RNG1D(rng))*(b - a)*delta;
            xi -= h1*0.5f;
            if (xi < 0.0f)
                return a + ((float)i + 0.5f + 0.5f*RNG1D(rng))*(b - a)*delta;
        }
        return 0.0f;
    }
    case TGHIP_TRANS_ERLANG: {
        if (!startOnSurface) {
            float x0 = RNG1D(rng), x1 = RNG1D(rng);
            return -1.0f/p0*logfH(x0*x1);
        }
        float xi = RNG1D(rng);
        float x = 0.5f;
        for (int i = 0; i < 10; ++i) {
            x += (xi - (1.0f - transLeafKernel(m, 0, x)))/transLeafKernel(m, 1, x);
            x = fmaxf(x, 0.0f);
        }
        return x;
    }
    case TGHIP_TRANS_DAVIS:
        return startOnSurface ? p0*(powfH(1.0f - RNG1D(rng), -1.0f/p0) - 1.0f) : p0*(powfH(1.0f - RNG1D(rng), -1.0f/(1.0f + p0)) - 1.0f);
    case TGHIP_TRANS_DAVIS_WEINSTEIN: {                 /* bisection on the cdf (:89-118) */
        float xi = RNG1D(rng);
        float step = 1e6f, result = step*2;
        while (step > 1e-6) {
            float cdf = 1.0f - transLeafKernel(m, startOnSurface ? 0 : 2, result);
            if (cdf > xi) result -= step; else result += step;
            step /= 2;
        }
        return result;
    }
    default:
        return -logfH(1.0f - RNG1D(rng));
    }
}

template<uint32_t M>
PT_DEV float transSample(const TgHipMedium &m, Rng &rng, bool startOnSurface)   /* InterpolatedTransmittance.cpp:65-72 */
{
    if (m.trans_type != TGHIP_TRANS_INTERPOLATED) return transLeafSample<M>(m, rng, startOnSurface);
    return rngNextBoolean(rng, m.trans_p[0]) ? transLeafSample<M>((&m)[2], rng, startOnSurface) : transLeafSample<M>((&m)[1], rng, startOnSurface);
}

/* HomogeneousMedium::sampleDistance (:66-107).  maxT = ray.farT(); false = the path ends here ("return emission").
 * MediumState::firstScatter (the "start on a surface" flag of the transmittance) is stateBounce == 0: reset() clears both,
 * advance() clears the flag and counts (Medium.hpp:36-46). */
/* The exponential and the atmospheric medium carry the exponential transmittance only (include/tungsten_hip.h; checked at upload): its four kernels
 * are all FastMath::exp(-tau), its sigmaBar is one, its two samplers -log(1 - xi) (ExponentialTransmittance.cpp:26-63) -- the general transEval /
 * transKernel3 / transSample above inline every transmittance type at every call site, which is 50 k instructions per medium type in the media kernels. */
PT_DEV f3 transExp3(f3 tau) { return mk3(fmathExp(-tau.x), fmathExp(-tau.y), fmathExp(-tau.z)); }
template<uint32_t M>
PT_DEV float transExpSample(Rng &rng) { return -logfH(1.0f - RNG1D(rng)); }
/* ExponentialMedium::densityIntegral / inverseOpticalDepth (ExponentialMedium.cpp:81-104): std::exp / std::log on floats = glibc's expf / logf */
PT_DEV float expMediumDensityIntegral(float x, float dx, float tMax)
{
    if (tMax == PT_INF)
        return expfH(-x)/dx;
    else if (dx == 0.0f)
        return expfH(-x)*tMax;
    else
        return (expfH(-x) - expfH(-dx*tMax - x))/dx;
}
PT_DEV float expMediumInverseOpticalDepth(float x, float dx, float tau)
{
    if (dx == 0.0f)
        return tau/expfH(-x);
    float denom = 1.0f - dx*expfH(x)*tau;
    return denom <= 0.0f ? PT_INF : -logfH(denom)/dx;
}
/* AtmosphericMedium (media/AtmosphericMedium.cpp): density exp(-s^2 (|p - center|^2 - radius^2)) with s = falloff_scale / radius -- in TgHipMedium:
 * falloff_scale = s (_effectiveFalloffScale), unit_point = _center, falloff_dir[0] = _radius.  The optical depth along a ray is a difference of error
 * functions: densityIntegral (:104-111) in float through Abramowitz & Stegun's 7.1.26 as math/Erf.hpp:247-283 writes it (std::exp on floats =
 * glibc's expf); inverseOpticalDepth (:113-122) in DOUBLE -- std::erf, std::exp, and Boost's erf_inv over std::log and std::sqrt: glibc's double erf /
 * exp / log restated (pt_libm.h: erfD, expD, logD -- matched against the host libm on 10^9 arguments each), the correctly rounded square root. */
/* (real calls, not inlined: the atmosphere's double-precision code would otherwise be copied into every call site of the media kernels) */
#define PT_DEV_CALL __device__ __attribute__((noinline))
#define PT_SQRT_PI     1.77245385091f           /* math/Angle.hpp:15-16 */
#define PT_INV_SQRT_PI (1.0f/PT_SQRT_PI)
PT_DEV float erfcAS(float x)                                   /* Erf::erfc<float> (math/Erf.hpp:249-263) */
{
    const float p = 0.32759f;
    const float as[5] = {0.254829592f, -0.284496736f, 1.421413741f, -1.453152027f, 1.061405429f};
    float t = 1.0f/(1.0f + p*fabsf(x));
    float ti = copysignf(t*expfH(-x*x), x);
    float result = 0.0f;
    for (int i = 0; i < 5; ++i) {
        result += as[i]*ti;
        ti *= t;
    }
    float constant = 1.0f - copysignf(1.0f, x);
    return constant + result;
}
PT_DEV float erfDifferenceAS(float x0, float x1)              /* Erf::erfDifference<float> (math/Erf.hpp:265-283) */
{
    const float p = 0.32759f;
    const float as[5] = {0.254829592f, -0.284496736f, 1.421413741f, -1.453152027f, 1.061405429f};
    float t0 = 1.0f/(1.0f + p*fabsf(x0));
    float t1 = 1.0f/(1.0f + p*fabsf(x1));
    float ti0 = copysignf(t0*expfH(-x0*x0), x0);
    float ti1 = copysignf(t1*expfH(-x1*x1), x1);
    float result = 0.0f;
    for (int i = 0; i < 5; ++i) {
        result += as[i]*(ti0 - ti1);
        ti0 *= t0;
        ti1 *= t1;
    }
    float constant = copysignf(1.0f, x1) - copysignf(1.0f, x0);
    return constant + result;
}
PT_DEV_CALL float atmDensityIntegral(float s, float radius, float h, float t0, float t1)   /* AtmosphericMedium::densityIntegral (:104-111) */
{
    if (t1 == PT_INF)
        return (PT_SQRT_PI*0.5f/s)*expfH((-h*h + radius*radius)*s*s)*erfcAS(s*t0);
    else
        return (PT_SQRT_PI*0.5f/s)*expfH((-h*h + radius*radius)*s*s)*erfDifferenceAS(s*t0, s*t1);
}
PT_DEV float atmDensity(float s, float radius, float h, float t0)                     /* AtmosphericMedium::density(h, t0) (:99-102) */
{
    return expfH(-(s*s)*(h*h - radius*radius + t0*t0));
}
template<int N>
PT_DEV double polyEvalD(double x, const double *P)             /* Polynomial::eval<Size> (math/Polynomial.hpp:9-18): Horner, no fused operations */
{
    double result = P[N - 1];
    for (int i = N - 2; i >= 0; --i) {
        result *= x;
        result += P[i];
    }
    return result;
}
PT_DEV double sqrtD(double x) { return __dsqrt_rn(x); }        /* std::sqrt on a double: IEEE, round to nearest */
PT_DEV double erfInvD(double z)                                 /* Erf::erfInv (math/Erf.hpp:192-245); tables: pt_erfinv_table.h */
{
    double p, q, sgn;
    if (z < 0) { p = -z; q = 1 - p; sgn = -1; }
    else       { p = z;  q = 1 - z; sgn = 1; }
    double result;
    if (p <= 0.5) {
        double g = p*(p + 10.0);
        double r = polyEvalD<8>(p, g_erfInvP1)/polyEvalD<10>(p, g_erfInvQ1);
        result = g*g_erfInvY[0] + g*r;
    } else if (q >= 0.25) {
        double g = sqrtD(-2.0*ptlibm::logD(q));
        double xs = q - 0.25;
        double r = polyEvalD<9>(xs, g_erfInvP2)/polyEvalD<9>(xs, g_erfInvQ2);
        result = g/(g_erfInvY[1] + r);
    } else {
        double x = sqrtD(-ptlibm::logD(q));
        if (x < 3.0) {
            double xs = x - 1.125;
            double R = polyEvalD<11>(xs, g_erfInvP3)/polyEvalD<8>(xs, g_erfInvQ3);
            result = g_erfInvY[2]*x + R*x;
        } else if (x < 6.0) {
            double xs = x - 3;
            double R = polyEvalD<9>(xs, g_erfInvP4)/polyEvalD<7>(xs, g_erfInvQ4);
            result = g_erfInvY[3]*x + R*x;
        } else if (x < 18.0) {
            double xs = x - 6.0;
            double R = polyEvalD<9>(xs, g_erfInvP5)/polyEvalD<7>(xs, g_erfInvQ5);
            result = g_erfInvY[4]*x + R*x;
        } else if (x < 44.0) {
            double xs = x - 18.0;
            double R = polyEvalD<8>(xs, g_erfInvP6)/polyEvalD<7>(xs, g_erfInvQ6);
            result = g_erfInvY[5]*x + R*x;
        } else {
            double xs = x - 44.0;
            double R = polyEvalD<8>(xs, g_erfInvP7)/polyEvalD<7>(xs, g_erfInvQ7);
            result = g_erfInvY[6]*x + R*x;
        }
    }
    return sgn*result;
}
PT_DEV_CALL float atmInverseOpticalDepth(float sF, float radiusF, double h, double t0, double tau)   /* AtmosphericMedium::inverseOpticalDepth (:113-122) */
{
    const double s = sF, radius = radiusF;
    const double inner = ptlibm::erfD(s*t0) + 2.0*double(PT_INV_SQRT_PI)*ptlibm::expD(s*s*(h - radius)*(h + radius))*s*tau;
    if (inner >= 1.0)
        return PT_INF;
    return float(erfInvD(inner)/s);
}
template<uint32_t M>
PT_DEV bool mediumSampleDistance(const DeviceScene &s, int medium, Rng &rng, f3 rayO, f3 rayD, float maxT, uint32_t &stateBounce, f3 &weight, float &t, bool &exited)
{
    const TgHipMedium &m = s.media[medium];
    if ((int)stateBounce > m.max_bounce)
        return false;
    const bool firstScatter = stateBounce == 0u;
    const f3 sigmaT = ld3(m.sigma_t);
    if (m.medium_type == TGHIP_MEDIUM_EXPONENTIAL) {             /* ExponentialMedium::sampleDistance (ExponentialMedium.cpp:106-150); exponential transmittance */
        const float x = m.falloff_scale*dot(rayO - ld3(m.unit_point), ld3(m.falloff_dir));
        const float dx = m.falloff_scale*dot(rayD, ld3(m.falloff_dir));
        if (m.absorption_only) {
            if (maxT == PT_INF && dx <= 0.0f)
                return false;
            t = maxT;
            weight = transExp3(sigmaT*expMediumDensityIntegral(x, dx, maxT));
            exited = true;
            return true;
        }
        int component = (int)(rngNext1D(rng)*3);
        float sigmaTc = component == 0 ? sigmaT.x : component == 1 ? sigmaT.y : sigmaT.z;
        float tauC = transExpSample<M>(rng)/sigmaTc;
        float tt = expMediumInverseOpticalDepth(x, dx, tauC);
        t = fminf(tt, maxT);
        exited = tt >= maxT;
        f3 tau = sigmaT*expMediumDensityIntegral(x, dx, t);
        weight = transExp3(tau);
        float pdf;
        if (exited) {
            pdf = avg3(transExp3(tau));
        } else {
            float rho = expfH(-(x + dx*t));
            pdf = avg3((sigmaT*rho)*transExp3(tau));
            weight = weight*((ld3(m.sigma_s)*rho)*1.0f);
        }
        weight = weight/pdf;
        stateBounce++;
        return true;
    }
    if (m.medium_type == TGHIP_MEDIUM_ATMOSPHERE) {              /* AtmosphericMedium::sampleDistance (AtmosphericMedium.cpp:124-168); exponential transmittance */
        const float sc = m.falloff_scale, radius = m.falloff_dir[0];
        const f3 p = rayO - ld3(m.unit_point);
        const float t0 = dot(p, rayD);
        const float h = length(p - rayD*t0);
        const float maxTa = maxT + t0;
        if (m.absorption_only) {
            t = maxT;
            weight = transExp3(sigmaT*atmDensityIntegral(sc, radius, h, t0, maxTa));
            exited = true;
            return true;
        }
        int component = (int)(rngNext1D(rng)*3);
        float sigmaTc = component == 0 ? sigmaT.x : component == 1 ? sigmaT.y : sigmaT.z;
        float tauC = transExpSample<M>(rng)/sigmaTc;
        float tt = atmInverseOpticalDepth(sc, radius, h, t0, tauC);
        t = fminf(tt, maxTa);
        exited = tt >= maxTa;
        f3 tau = sigmaT*atmDensityIntegral(sc, radius, h, t0, t);
        weight = transExp3(tau);
        float pdf;
        if (exited) {
            pdf = avg3(transExp3(tau));
        } else {
            float rho = atmDensity(sc, radius, h, t);
            pdf = avg3((sigmaT*rho)*transExp3(tau));
            weight = weight*((ld3(m.sigma_s)*rho)*1.0f);
        }
        weight = weight/pdf;
        t -= t0;
        stateBounce++;
        return true;
    }
    if (m.trans_type == TGHIP_TRANS_EXPONENTIAL) {               /* the usual medium, without the transmittance switch at five call sites (same arithmetic) */
        if (m.absorption_only) {
            if (maxT == PT_INF)
                return false;
            t = maxT;
            weight = transExp3(sigmaT*t);
            exited = true;
            return true;
        }
        int component = (int)(rngNext1D(rng)*3);
        float sigmaTc = component == 0 ? sigmaT.x : component == 1 ? sigmaT.y : sigmaT.z;
        float tt = transExpSample<M>(rng)/sigmaTc;
        t = fminf(tt, maxT);
        exited = tt >= maxT;
        f3 tau = sigmaT*t;
        weight = transExp3(tau);
        float pdf;
        if (exited) {
            pdf = avg3(transExp3(tau));
        } else {
            pdf = avg3(sigmaT*transExp3(tau));
            weight = weight*(ld3(m.sigma_s)*1.0f);
        }
        weight = weight/pdf;
        stateBounce++;
        return true;
    }
    if (m.absorption_only) {
        if (maxT == PT_INF)
            return false;
        t = maxT;
        weight = transEval(m, sigmaT*t, firstScatter, true);
        exited = true;
    } else {
        int component = (int)(rngNext1D(rng)*3);                     /* sampler.nextDiscrete(3): the supplemental stream */
        float sigmaTc = component == 0 ? sigmaT.x : component == 1 ? sigmaT.y : sigmaT.z;
        float tt = transSample<M>(m, rng, firstScatter)/sigmaTc;
        t = fminf(tt, maxT);
        exited = tt >= maxT;
        f3 tau = sigmaT*t;
        weight = transEval(m, tau, firstScatter, exited);
        float pdf;
        if (exited) {
            pdf = avg3(transKernel3(m, firstScatter ? 0 : 2, tau));            /* surfaceProbability */
        } else {
            pdf = avg3(sigmaT*transKernel3(m, firstScatter ? 1 : 3, tau));     /* sigmaT*mediumPdf */
            weight = weight*(ld3(m.sigma_s)*transSigmaBar(m));
        }
        weight = weight/pdf;
        stateBounce++;                                               /* state.advance() */
    }
    return true;
}

PT_DEV f3 mediumTransmittance(const DeviceScene &s, int medium, f3 rayO, f3 rayD, float farT, bool startOnSurface, bool endOnSurface)   /* HomogeneousMedium::transmittance (:109-116) */
{
    const TgHipMedium &m = s.media[medium];
    if (m.medium_type == TGHIP_MEDIUM_EXPONENTIAL) {             /* ExponentialMedium::transmittance (ExponentialMedium.cpp:151-163) */
        const float x = m.falloff_scale*dot(rayO - ld3(m.unit_point), ld3(m.falloff_dir));
        const float dx = m.falloff_scale*dot(rayD, ld3(m.falloff_dir));
        if (farT == PT_INF && dx <= 0.0f)
            return splat3(0.0f);
        return transExp3(ld3(m.sigma_t)*expMediumDensityIntegral(x, dx, farT));
    }
    if (m.medium_type == TGHIP_MEDIUM_ATMOSPHERE) {              /* AtmosphericMedium::transmittance (AtmosphericMedium.cpp:170-180) */
        const f3 p = rayO - ld3(m.unit_point);
        const float t0 = dot(p, rayD);
        const float t1 = farT + t0;
        const float h = length(p - rayD*t0);
        return transExp3(ld3(m.sigma_t)*atmDensityIntegral(m.falloff_scale, m.falloff_dir[0], h, t0, t1));
    }
    if (farT == PT_INF)
        return splat3(0.0f);
    if (m.trans_type == TGHIP_TRANS_EXPONENTIAL)
        return transExp3(ld3(m.sigma_t)*farT);
    return transEval(m, ld3(m.sigma_t)*farT, startOnSurface, endOnSurface);
}

/* PhaseFunction::eval == pdf (IsotropicPhaseFunction.cpp:17-41, HenyeyGreensteinPhaseFunction.cpp:16-43, 80-83) */
PT_DEV float phaseHG(float g, float cosTheta)
{
    float term = 1.0f + g*g - 2.0f*g*cosTheta;
    return PT_INV_FOUR_PI*(1.0f - g*g)/(term*sqrtf(term));
}
PT_DEV float phaseRayleigh(float cosTheta) { return (3.0f/(16.0f*PT_PI))*(1.0f + cosTheta*cosTheta); }   /* RayleighPhaseFunction.cpp:14-17 */
PT_DEV float phaseEval(const TgHipMedium &m, f3 wi, f3 wo)
{
    if (m.phase_type == TGHIP_PHASE_RAYLEIGH) return phaseRayleigh(dot(wi, wo));
    return m.phase_type == TGHIP_PHASE_HENYEY_GREENSTEIN ? phaseHG(m.phase_g, dot(wi, wo)) : PT_INV_FOUR_PI;
}
template<uint32_t M>
PT_DEV void phaseSample(const TgHipMedium &m, Rng &rng, f3 wi, f3 &w, float &pdf)   /* sample.weight = 1 */
{
    float xi0 = RNG1D(rng), xi1 = RNG1D(rng);
    const float g = m.phase_g;
    if (m.phase_type == TGHIP_PHASE_RAYLEIGH) {                      /* RayleighPhaseFunction::sample (:31-49) */
        float phi = xi0*PT_TWO_PI;
        float z = xi1*4.0f - 2.0f;
        float invZ = sqrtf(z*z + 1.0f);
        float u = cbrtfH(z + invZ);
        float cosTheta = u - 1.0f/u;
        float sinTheta = sqrtf(fmaxf(1.0f - cosTheta*cosTheta, 0.0f));
        Frame f = frameFromNormal(wi);
        float sinPhi, cosPhi;
        sincosfH(phi, sinPhi, cosPhi);
        w = toGlobal(f, mk3(cosPhi*sinTheta, sinPhi*sinTheta, cosTheta));
        pdf = phaseRayleigh(cosTheta);
    } else if (m.phase_type != TGHIP_PHASE_HENYEY_GREENSTEIN || g == 0.0f) {
        w = uniformSphere(xi0, xi1);
        pdf = PT_INV_FOUR_PI;
    } else {
        float phi = xi0*PT_TWO_PI;
        float cosTheta = (1.0f + g*g - sqr((1.0f - g*g)/(1.0f + g*(xi1*2.0f - 1.0f))))/(2.0f*g);
        float sinTheta = sqrtf(fmaxf(1.0f - cosTheta*cosTheta, 0.0f));
        Frame f = frameFromNormal(wi);
        float sinPhi, cosPhi;
        sincosfH(phi, sinPhi, cosPhi);
        w = toGlobal(f, mk3(cosPhi*sinTheta, sinPhi*sinTheta, cosTheta));
        pdf = phaseHG(g, cosTheta);
    }
}

#endif
