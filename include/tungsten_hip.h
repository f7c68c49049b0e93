/*
 * tungsten_hip.h -- C-ABI boundary of the MI355X-native `path_tracer_hip` integrator.
 *
 * This is the thin extern "C" shim the north star asks for: plain pointers and sizes,
 * no C++/torch types.  It is what a Tungsten maintainer would bind from the reference's
 * C++11 host.  Each entry point names the reference interface it
 * replaces (paths relative to the reference tree, src/core/...):
 *
 *   tghip_upload_scene      <- TraceableScene::TraceableScene (renderer/TraceableScene.hpp:57-137):
 *                              the Embree top-level + per-mesh scenes (primitives/TriangleMesh.cpp:524-572)
 *                              become one flattened BVH2 + SoA record stream in HBM.
 *   tghip_render_pass/_wait <- PathTraceIntegrator::startRender / waitForCompletion
 *                              (integrators/path_tracer/PathTraceIntegrator.cpp:220-244) and, inside them,
 *                              renderTile (:136-156) -> PathTracer::traceSample (PathTracer.cpp:14-149).
 *   tghip_abort             <- PathTraceIntegrator::abortRender (PathTraceIntegrator.cpp:246-256).
 *   tghip_download_framebuffer <- OutputBuffer<Vec3f>::addSample / operator[] (cameras/OutputBuffer.hpp:104-144).
 *   tghip_trace_rays        <- TraceableScene::intersect (renderer/TraceableScene.hpp:170-192), batched.
 *   tghip_reduce_framebuffers <- the merge of per-machine partial renders the reference does offline on the host
 *                              (src/hdrmanip/hdrmanip.cpp:69-112: load N HDR images, add, divide); here the tile shards of
 *                              one frame, summed on the device side by RCCL over xGMI.
 *   tghip_comm_unique_id / tghip_comm_init_rank / tghip_reduce_framebuffer_rank <- the same merge with one PROCESS per GPU (hdrmanip.cpp:69-112
 *                              again: there the partial renders come from separate `tungsten` processes).
 *   tghip_get_counters      <- (no reference analogue; feeds the roofline model).
 *   tghip_get_walk_stats    <- (no reference analogue; lane utilisation of the walks).
 *   tghip_debug_libm        <- std::sin / cos / log / exp / acos / atan2 / pow / cbrt on floats as the reference's path calls them (glibc's sinf ... cbrtf):
 *                              the device's restatements evaluated on the device, for the parity tests.
 *
 * Ownership: the caller owns every host array (borrowed for the duration of the call; the
 * shim copies with hipMemcpyAsync); the shim owns device memory behind the opaque handle.
 * All functions return 0 on success and a negative TGHIP_E_* code on failure (message via
 * tghip_last_error); nothing throws across the boundary.  One handle per device; calls on
 * one handle must be serialised by the caller; different handles may be driven from
 * different host threads.
 *
 * The flattened scene structs below are also the input of the CPU oracle (oracle/oracle.c),
 * so that the checker and the HIP path consume bit-identical inputs.
 */
#ifndef TUNGSTEN_HIP_H_
#define TUNGSTEN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGHIP_ABI_VERSION 10

/* ---- error codes ------------------------------------------------------------------ */
enum {
    TGHIP_OK            =  0,
    TGHIP_E_INVALID     = -1,  /* bad argument / inconsistent description */
    TGHIP_E_NODEVICE    = -2,  /* no HIP device / hipSetDevice failed     */
    TGHIP_E_HIP         = -3,  /* a HIP runtime call failed               */
    TGHIP_E_NOSCENE     = -4,  /* render/trace before upload              */
    TGHIP_E_ABORTED     = -5,  /* pass was aborted via tghip_abort        */
    TGHIP_E_UNSUPPORTED = -6   /* feature outside the hot-path scope      */
};

/* ---- flattened geometry -------------------------------------------------------------
 * One BVH2 over all finite primitives in world space (the reference bakes transforms into
 * _tfVerts, TriangleMesh.cpp:542-552).  64-byte nodes; a child reference is either an
 * internal node index (>= 0) or a leaf (bit 31 set): count = (ref >> 27) & 15, first record
 * = ref & 0x07FFFFFF.  Leaves are contiguous runs of 48-byte primitive records. */
/* a node of the reference's own BVH over the instances of an `instances` primitive as the host's restatement of its builder
 * produces it (bvh/BinaryBvh.hpp: TinyBvhNode): box = x: lmin rmin lmax rmax, then y, then z; count = 0: inner node, children
 * left and left + 1; else a leaf of `count` (1 or 2) instances at slots [left, left + count) of BinaryBvh::_primIndices.
 * Host-side only: the scene description carries these trees as TgHipBvhNode entries (see the instance-set record). */
typedef struct TgHipInstNode {
    float box[12];
    uint32_t left, count;
    uint32_t pad[2];
} TgHipInstNode;              /* 64 B */

typedef struct TgHipBvhNode {
    float   lo0[3], hi0[3];   /* child 0 bounds */
    float   lo1[3], hi1[3];   /* child 1 bounds */
    int32_t child0, child1;
    uint32_t pad[2];
} TgHipBvhNode;               /* 64 B */

#define TGHIP_LEAF_FLAG        0x80000000u
#define TGHIP_LEAF_COUNT(ref)  (((uint32_t)(ref) >> 27) & 15u)
#define TGHIP_LEAF_FIRST(ref)  ((uint32_t)(ref) & 0x07FFFFFFu)
#define TGHIP_MAKE_LEAF(first, count) (int32_t)(TGHIP_LEAF_FLAG | ((uint32_t)(count) << 27) | (uint32_t)(first))
#define TGHIP_MAX_LEAF         15
#define TGHIP_MAX_TREE_DEPTH   48   /* the builder's bound for ONE tree */
#define TGHIP_MAX_BVH_DEPTH    96   /* builder guarantees depth <= this (device stack size; single-level trees stay below 48, the
                                       three-level walk of scenes with instances adds the levels of its trees) */
/* Scenes with at most this many primitive records are intersected as a flat list in record order (every ray
 * tests every record; the wave walks the list uniformly, so record data comes through the scalar cache) instead of
 * through the BVH -- the analogue of the reference's top-level Embree scene over a handful of user-geometry
 * primitives (TraceableScene.hpp:112-134).  The oracle follows the same rule so that visit counts agree.
 * Flat lists with TgHipSceneDesc::top_nodes (scenes whose records are all quads, cubes, spheres, disks or cylinders): the closest hit is the one the reference's
 * Embree walk returns where faces coincide -- see TgHipTopNode below. */
#define TGHIP_FLAT_MAX_RECS    16

/* ---- the reference's top-level tree ----------------------------------------------------------------------------------------------
 * TraceableScene commits ONE Embree user geometry whose items are the scene's finite primitives (renderer/TraceableScene.hpp:112-134) and
 * Embree builds a BVH4 with one item per leaf over their bounds().  Where faces coincide -- a block standing ON the floor quad, a light lying
 * IN the ceiling, a ray into the seam of two walls -- the ORDER in which a ray visits that tree decides which primitive it hits
 * (kernels/bvh/bvh_intersector1.cpp:60-125, bvh_traverser1.h:41-104):
 *   - a child counts only if the ray passes its box by the node's slab test (rdir = rcp(dir) by RCPPS + one Newton step, planes
 *     (bound - org)*rdir, max / min and the comparison on the bit patterns as signed integers), under the hit distance found SO FAR;
 *   - of the children hit the nearest box entry is descended into first, the others wait on a stack sorted by entry (one hit child: go; two:
 *     the first only `if (d0 < d1)`; three / four: Embree's sorting networks, stack_item.h:44-60);
 *   - a child popped behind the hit found so far is skipped -- by its BOX's entry distance, which can lie an ulp behind the distance the
 *     primitive's own intersect() reports; a quad accepts t <= farT, a cube or sphere t < farT.
 * The tree is therefore part of the path's arithmetic.  csrc/host/EmbreeTopTree.cpp restates Embree 2.11's builder for it (binned SAH, four
 * children, leaf size one) node for node; tgh_top_tree_build (tungsten_host.h) produces it, tests/test_top_tree.py holds it to trees read out of
 * the reference's own Embree.  Scenes that carry it: flat lists of quads, cubes, spheres, disks and cylinders (one item per object that has a record, in
 * object order); every other scene passes NULL / 0 and is walked as before (a triangle mesh lives in ONE tree with the other records there).
 * Node 0 is the root, nodes in preorder.  child[i] >= 0: a node; < 0: the record ~child[i]; TGHIP_TOP_EMPTY: unused slot (its box is
 * lower = +inf, upper = -inf, as Embree clears it). */
#define TGHIP_TOP_EMPTY  0x7FFFFFFF
#define TGHIP_TOP_MAX_DEPTH 7       /* levels of nodes; Embree's tree over <= TGHIP_FLAT_MAX_RECS items has at most 5 (a node has four children or leaves only) */
typedef struct TgHipTopNode {
    float   lower[4][3];
    float   upper[4][3];
    int32_t child[4];
} TgHipTopNode;            /* 112 bytes */

/* ---- 8-wide BVH with quantised child boxes ------------------------------------------------------------------------
 * What the traversal kernels of single-level scenes walk (the BVH2 above stays the structure of the two-level
 * instance walk and of the transparency / media shadow walks).  It is the BVH2 collapsed: every wide node stands for a
 * BVH2 node, its up to eight children for BVH2 descendants (the child with the largest box is opened until the node is
 * full), every leaf child for ONE BVH2 leaf.  A node is 80 bytes = five 16-byte loads of ONE lane:
 *   child box plane k of axis a = origin[a] + q * 2^(exp[a] - 127)           (q in 0..255; lower planes rounded down,
 *                                                                              upper planes up: boxes only ever grow)
 *   slot s holds an internal child iff imask bit s; internal children are consecutive nodes from child_base in slot order
 *   leaf_valid bit 4 s + j: the leaf child in slot s has a j-th record (j < its record count <= 4); that record is
 *     recs[rec_base + number of leaf_valid bits below it]: the leaf children of one node share one contiguous run of at
 *     most 32 records, in slot order (records are ordered by wide node, breadth first)
 *   an empty slot has qlo = 255 > qhi = 0 on every axis (no ray passes its box test) and no leaf_valid bits
 * Children sit in the slot whose sign pattern (bit 0: +x, bit 1: +y, bit 2: +z) best matches the direction from the
 * node's centre to theirs, so that visiting hit slots in ascending (slot XOR ray octant) order is roughly front to back
 * without sorting distances (Ylitie, Karras, Laine: "Efficient incoherent ray traversal on GPUs through compressed wide
 * BVHs", HPG 2017 -- the node layout here is this repository's own).  Replaces what Embree's BVH4 does for the
 * reference (thirdparty/embree/kernels/bvh/bvh_intersector1.cpp:48-142). */
typedef struct TgHipWideNode {
    float    origin[3];
    uint8_t  exp[3];          /* biased exponents of the per-axis plane spacing */
    uint8_t  imask;
    uint32_t child_base;
    uint32_t rec_base;
    uint32_t leaf_valid;
    uint32_t reserved;
    uint8_t  qlo[3][8];       /* [axis][slot] */
    uint8_t  qhi[3][8];
} TgHipWideNode;              /* 80 B */
#define TGHIP_WIDE_MAX_LEAF   4    /* records per leaf child */
#define TGHIP_MAX_WIDE_DEPTH  32   /* builder guarantees depth <= this (device stack: one 8-byte entry per level) */

/* record kinds (meta >> 29) */
enum { TGHIP_REC_TRIANGLE = 0, TGHIP_REC_QUAD = 1, TGHIP_REC_CUBE = 2, TGHIP_REC_SPHERE = 3, TGHIP_REC_INSTANCE = 4,
       TGHIP_REC_DISK = 5, TGHIP_REC_CYLINDER = 6, TGHIP_REC_INSTANCE_SET = 7 };
#define TGHIP_REC_KIND(meta)   ((uint32_t)(meta) >> 29)
#define TGHIP_REC_OBJECT(meta) ((uint32_t)(meta) & 0x1FFFFFFFu)

/* 48-byte primitive record = three float4:
 *   triangle: a = v0, b = v1 - v0, c = v2 - v0                (p0,p1 unused)
 *   quad    : a = base, b = edge0, c = edge1, p0/p1 = 1/|edge0|^2, 1/|edge1|^2   (Quad.cpp:298-316)
 *   cube / sphere / disk / cylinder: geometry lives in objects[TGHIP_REC_OBJECT(meta)]; a,b,c unused
 *   instance: one rigid placement of a master mesh (primitives/Instance.cpp:290-344): a = _instancePos[i],
 *             (p0, b) = _instanceRot[i] as quaternion (w; x, y, z), c[0] = bits of the master's BVH root node index
 *             (uint32), c[1] = bits of the instance number i, c[2] = bits of the root node index of the master's wide
 *             subtree (uint32; scenes with a wide BVH); the object is the `instances` primitive.  The master's
 *             triangle records (in master space, i.e. with the master's own transform applied) and its BVH2 subtree
 *             follow the top-level ones in recs / tri_attrs / nodes and are reachable only through instance records.
 *             (ABI 8: c[1] = bits of the first inst_prims slot of the instance's leaf in the reference's tree: its box is inst_leaf_boxes[8 c[1]].)
 *   instance set (ABI 8): one `instances` primitive as the reference intersects it -- Instance::intersect walks ITS OWN BVH over
 *             the instances (bvh/BinaryBvh.hpp) and keeps the LAST hit in that tree's visiting order, each instance having
 *             been handed a ray with farT = infinity (primitives/Instance.cpp:290-311).  a = min, b = max of the primitive's
 *             bounds (BinaryBvh::_bounds), c[0] = bits of the root of the reference's tree, restated node for node by the host
 *             (csrc/host/RefInstanceBvh.cpp) and stored as ordinary TgHipBvhNode entries with the reference's exact child
 *             boxes: a node index, or -- one or two instances in all -- a leaf reference.  Leaf references of THAT tree
 *             index inst_prims[] (BinaryBvh::_primIndices: the instance records, kind TGHIP_REC_INSTANCE, of the leaf's one
 *             or two instances).  The closest-hit walk reaches the set through the scene's BVH2 (nodes[0]: over the
 *             non-instance records and the sets); the wide BVH -- walked by any-hit shadow queries, for which the order is
 *             immaterial -- holds the instance records themselves, built from their tight geometry boxes (inst_tight_boxes, below);
 *             the box of the instance's leaf in the reference's tree (inst_leaf_boxes) is tested when the record is reached. */
typedef struct TgHipPrimRec {
    float a[3]; uint32_t meta;
    float b[3]; float p0;
    float c[3]; float p1;
} TgHipPrimRec;               /* 48 B */

/* 64-byte per-triangle shading attributes, same index as the record (denormalised from the
 * reference's Vertex{pos,normal,uv}/TriangleI{v0,v1,v2,material}, primitives/Vertex.hpp:10-13,
 * Triangle.hpp:14-28, so that one hit costs one 64-B gather).  Normals are already multiplied
 * by the normal matrix (TriangleMesh.cpp:542-549). */
typedef struct TgHipTriAttr {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    int32_t bsdf;             /* global bsdf index = mesh._bsdfs[clamp(tri.material)] */
} TgHipTriAttr;               /* 64 B */

/* ---- objects (one per reference Primitive) ------------------------------------------ */
enum { TGHIP_OBJ_MESH = 0, TGHIP_OBJ_QUAD = 1, TGHIP_OBJ_CUBE = 2, TGHIP_OBJ_SPHERE = 3,
       TGHIP_OBJ_INFINITE_SPHERE = 4, TGHIP_OBJ_INSTANCES = 5, TGHIP_OBJ_DISK = 6,
       TGHIP_OBJ_INFINITE_SPHERE_CAP = 7,     /* sun-like emitter: normal = _capDir, scale[0] = _cosCapAngle, edge0/edge1 = _capFrame tangent/bitangent (InfiniteSphereCap.cpp:233-249) */
       TGHIP_OBJ_POINT = 8,
       TGHIP_OBJ_CYLINDER = 9 };              /* primitives/Cylinder.cpp:305-319: pos = _pos, rot = _rot, normal = _axis, scale = {_radius, _halfHeight, _capped ? 1 : 0} */                 /* Dirac point light (primitives/Point.cpp): pos = _pos, scale = _power as Point.cpp:186 leaves it; never hit, sampled without random numbers */
#define TGHIP_OBJF_SMOOTH   1u   /* mesh "smooth": Ns interpolated (TriangleMesh.cpp:344-355) */
#define TGHIP_OBJF_SAMPLE   2u   /* infinite_sphere "sample" (InfiniteSphere.cpp:117-122)      */
#define TGHIP_OBJF_SKYDOME  4u   /* a TGHIP_OBJ_INFINITE_SPHERE that is the `skydome` primitive (primitives/Skydome.cpp): its emission is the
                                    512 x 256 sky image baked at prepareForRender (:279-306); directions map to the image without the
                                    primitive's rotation (:41-60) and chooseLight weighs it with 4 pi instead of 2 pi (:240-243) */

typedef struct TgHipObject {
    int32_t  type;            /* TGHIP_OBJ_*                                      */
    int32_t  bsdf;            /* bsdf index (quad/cube/sphere); -1 for infinite   */
    int32_t  emission;        /* texture index of the radiance, -1 = not emissive */
    int32_t  light;           /* index in lights[] if samplable emitter, else -1  */
    uint32_t flags;
    float    area, inv_area;  /* quad/cube/mesh total area                        */
    int32_t  first_light_tri; /* sampled mesh emitters: float offset of the mesh's block in light_tris, else -1 */
    float    base[3], edge0[3], edge1[3], normal[3]; /* quad (Quad.cpp:298-316)    */
    float    inv_uv_sq[2];
    float    pos[3], scale[3];                       /* cube half-extent / sphere radius in scale[0] /
                                                        disk (Disk.cpp:303-315): pos = _center, normal = _n, scale = {_r, _cosApex, -}, edge0 = _frame.tangent,
                                                        edge1 = _frame.bitangent, base = _coneBase */
    float    rot[9];                                 /* row-major 3x3 rotation (cube; infinite sphere _rotTransform) */
    float    face_cdf[3];                            /* cube (Cube.cpp:353-370)    */
    int32_t  num_light_tris;  /* sampled mesh emitters: triangles in the block (TriangleMesh::makeSamplable) */
    int32_t  int_medium, ext_medium;  /* Primitive::_intMedium/_extMedium (Primitive.cpp:30-31) as indices into media[], -1 = none;
                                         the primitive overrides the path's medium iff either is set (Primitive.hpp:172-183) */
} TgHipObject;

/* ---- BSDFs ---------------------------------------------------------------------------- */
enum { TGHIP_BSDF_LAMBERT = 0, TGHIP_BSDF_NULL = 1, TGHIP_BSDF_ROUGH_CONDUCTOR = 2,
       TGHIP_BSDF_SMOOTH_COAT = 3, TGHIP_BSDF_DIELECTRIC = 4, TGHIP_BSDF_ROUGH_DIELECTRIC = 5,
       TGHIP_BSDF_MIRROR = 6, TGHIP_BSDF_CONDUCTOR = 7, TGHIP_BSDF_PLASTIC = 8,
       TGHIP_BSDF_ROUGH_PLASTIC = 9, TGHIP_BSDF_MIXED = 10, TGHIP_BSDF_TRANSPARENCY = 11,
       TGHIP_BSDF_FORWARD = 12, TGHIP_BSDF_ERROR = 13,
       /* ABI 9: the five remaining non-fibre types of bsdfs/BsdfFactory.cpp:29-51.  Their own parameters:
        *   DIFFUSE_TRANSMISSION  eta[0] = _transmittance (0.5: the reference has no JSON key for it)
        *   PHONG                 eta[0] = _exponent, eta[1] = _diffuseRatio, k[0] = _invExponent, k[1] = _pdfFactor, k[2] = _brdfFactor (PhongBsdf.cpp:126-132)
        *   THINSHEET             ior, tex1 = _thickness (scalar texture), sigma_a, enable_refraction = _enableInterference
        *   OREN_NAYAR            roughness (scalar texture)
        *   ROUGH_COAT            ior, thickness, sigma_a, scaled_sigma_a, avg_transmittance, distribution, roughness, sub0 = _substrate */
       TGHIP_BSDF_DIFFUSE_TRANSMISSION = 14, TGHIP_BSDF_PHONG = 15, TGHIP_BSDF_THINSHEET = 16, TGHIP_BSDF_OREN_NAYAR = 17,
       TGHIP_BSDF_ROUGH_COAT = 18 };
enum { TGHIP_DIST_BECKMANN = 0, TGHIP_DIST_PHONG = 1, TGHIP_DIST_GGX = 2 };
/* lobe bits, identical to bsdfs/BsdfLobes.hpp:13-33 */
enum { TGHIP_LOBE_GLOSSY_R = 1, TGHIP_LOBE_GLOSSY_T = 2, TGHIP_LOBE_DIFFUSE_R = 4, TGHIP_LOBE_DIFFUSE_T = 8,
       TGHIP_LOBE_SPECULAR_R = 16, TGHIP_LOBE_SPECULAR_T = 32, TGHIP_LOBE_ANISOTROPIC = 64,
       TGHIP_LOBE_FORWARD = 128 };

typedef struct TgHipBsdf {
    int32_t  type;            /* TGHIP_BSDF_*                               */
    uint32_t lobes;           /* after prepareForRender                      */
    int32_t  albedo;          /* texture index                               */
    int32_t  distribution;    /* TGHIP_DIST_*                                */
    int32_t  roughness;       /* texture index (scalar)                      */
    int32_t  sub0, sub1;      /* substrate / bsdf0,bsdf1 / base              */
    int32_t  tex1;            /* ratio / alpha texture                       */
    float    ior, thickness, avg_transmittance, diffuse_fresnel;
    int32_t  enable_refraction;
    float    eta[3], k[3], sigma_a[3], scaled_sigma_a[3];
    int32_t  bump1;           /* Bsdf::_bump (bsdfs/Bsdf.cpp:19-25, scalar request): texture index + 1 of a NON-constant bump map, 0 = none
                                 (a constant one changes nothing, Primitive.cpp:128-131); the shading frame then comes from the primitive's
                                 tangent space and the map's derivatives (Primitive::setupTangentFrame, Primitive.cpp:125-163).  (In what
                                 was padding up to ABI 7: a description written before carries 0.) */
    float    pad;
} TgHipBsdf;

/* ---- participating media (media/HomogeneousMedium.cpp) ------------------------------------------------- */
enum { TGHIP_PHASE_ISOTROPIC = 0, TGHIP_PHASE_HENYEY_GREENSTEIN = 1, TGHIP_PHASE_RAYLEIGH = 2 };   /* phasefunctions/{Isotropic,HenyeyGreenstein,Rayleigh}PhaseFunction.cpp */
/* transmittances/{Exponential,Linear,Quadratic,DoubleExponential,Pulse,Erlang,Davis,DavisWeinstein}Transmittance.cpp */
enum { TGHIP_TRANS_EXPONENTIAL = 0, TGHIP_TRANS_LINEAR = 1, TGHIP_TRANS_QUADRATIC = 2, TGHIP_TRANS_DOUBLE_EXPONENTIAL = 3,
       TGHIP_TRANS_PULSE = 4, TGHIP_TRANS_ERLANG = 5, TGHIP_TRANS_DAVIS = 6, TGHIP_TRANS_DAVIS_WEINSTEIN = 7,
       TGHIP_TRANS_INTERPOLATED = 8 };      /* InterpolatedTransmittance.cpp: trans_p = {ratio}; its two operands _trA, _trB are the trans_type / trans_p of the
                                               two media[] entries that FOLLOW this one (placeholders no primitive refers to; neither may be interpolated itself) */
typedef struct TgHipMedium {
    float   sigma_a[3], sigma_s[3], sigma_t[3];   /* after prepareForRender: material sigma x density (HomogeneousMedium.cpp:43-49) */
    int32_t absorption_only;                      /* _sigmaS == 0 */
    int32_t max_bounce;                           /* Medium::_maxBounce (Medium.cpp:16, 29) */
    int32_t phase_type;                           /* TGHIP_PHASE_* */
    float   phase_g;                              /* Henyey-Greenstein asymmetry */
    int32_t trans_type;                           /* TGHIP_TRANS_*: Medium::_transmittance (Medium.cpp:14, 27-28) */
    float   trans_p[3];                           /* linear / quadratic: {max_t}; double_exponential: {sigma_a, sigma_b}; pulse: {min, max, num_pulses};
                                                     erlang: {rate}; davis: {alpha}; davis_weinstein: {h, c} */
    /* ABI 9: media/ExponentialMedium.cpp next to the homogeneous medium -- the density falls off as exp(-falloff_scale (p - unit_point) . falloff_dir);
       sigma_* are the medium's coefficients at density one; exponential transmittance only (ExponentialMedium::sampleDistance hands the
       transmittance's eval an `exited` flag it has not set yet, :122-123: only the exponential one does not look at it) */
    int32_t medium_type;                          /* TGHIP_MEDIUM_* */
    float   falloff_scale;                        /* _falloffScale */
    float   unit_point[3];                        /* _unitPoint */
    float   falloff_dir[3];                       /* _unitFalloffDirection (normalised at prepareForRender) */
    float   pad[3];
} TgHipMedium;                /* 112 B */
/* media/AtmosphericMedium.cpp (TGHIP_MEDIUM_ATMOSPHERE): density exp(-s^2 (|p - center|^2 - radius^2)); the fields above are reused --
   falloff_scale = s = _effectiveFalloffScale (falloff_scale / radius after prepareForRender, :79), unit_point = _center (the pivot primitive's
   origin when the scene names one, :70-77), falloff_dir[0] = _radius; exponential transmittance only, for the reason given for the exponential
   medium (AtmosphericMedium::sampleDistance, :150-151) */
enum { TGHIP_MEDIUM_HOMOGENEOUS = 0, TGHIP_MEDIUM_EXPONENTIAL = 1, TGHIP_MEDIUM_ATMOSPHERE = 2 };

/* ---- textures ------------------------------------------------------------------------- */
enum { TGHIP_TEX_CONSTANT = 0, TGHIP_TEX_CHECKER = 1, TGHIP_TEX_BITMAP = 2 };
#define TGHIP_TEXF_LINEAR 1u
#define TGHIP_TEXF_CLAMP  2u
#define TGHIP_TEXF_RGB    4u
#define TGHIP_TEXF_VALID  8u

typedef struct TgHipTexture {
    int32_t  type;
    uint32_t flags;
    int32_t  w, h;
    float    value[3];        /* constant value                              */
    float    scale;           /* bitmap _scale                               */
    float    on_color[3];  int32_t res_u;
    float    off_color[3]; int32_t res_v;
    float    avg[3];          /* Texture::average()                          */
    float    pad;
    int64_t  texel_offset;    /* float offset into texels (3 per texel if RGB, else 1) */
    int64_t  dist_offset;     /* float offset into dist: marginalPdf[h] marginalCdf[h+1] pdf[w*h] cdf[(w+1)*h]; -1 = none
                                 (sampling/Distribution2D.hpp:18-83 built by BitmapTexture::makeSamplable, BitmapTexture.cpp:400-431) */
} TgHipTexture;

/* ---- camera (cameras/PinholeCamera.cpp:28-86, Camera.cpp:44-68, ReconstructionFilter) ---- */
enum { TGHIP_FILTER_DIRAC = 0, TGHIP_FILTER_BOX = 1, TGHIP_FILTER_TABULATED = 2 };
enum { TGHIP_CAMERA_PINHOLE = 0, TGHIP_CAMERA_THINLENS = 1,     /* cameras/PinholeCamera.cpp, cameras/ThinlensCamera.cpp */
       TGHIP_CAMERA_EQUIRECTANGULAR = 2, TGHIP_CAMERA_CUBEMAP = 3 };   /* cameras/CubemapCamera.cpp: six 90-degree faces laid out as a cross, a row or a column; inv_xf as for the
                                                                    equirectangular camera, blade_count = CubemapCamera::ProjectionMode (0 horizontal_cross, 1 vertical_cross, 2 row, 3 column) */                       /* cameras/EquirectangularCamera.cpp: the full sphere around pos, longitude across the image, latitude down it;
                                                                    inv_xf[0..8] then holds _rot = _transform.extractRotation() row-major (EquirectangularCamera.cpp:129-134) and
                                                                    inv_xf[9] = 1 / res_y (Camera::_pixelSize.y; pixel_size_x = 1 / res_x as for every camera) */
enum { TGHIP_APERTURE_DISK = 0, TGHIP_APERTURE_BLADE = 1,       /* textures/DiskTexture.cpp:78-81, textures/BladeTexture.cpp:110-130 */
       TGHIP_APERTURE_BITMAP = 2 };                             /* textures/BitmapTexture.cpp:433-439 with the MAP_UNIFORM distribution (:400-431) */
typedef struct TgHipCamera {
    float   pos[3];
    float   plane_dist;
    float   xf[9];            /* row-major upper 3x3 of Camera::_transform (right axis negated) */
    float   ratio, pixel_size_x;
    int32_t res_x, res_y;
    int32_t filter_type;
    float   filter_width, filter_bin_size;
    float   filter_cdf[32];   /* ReconstructionFilter::_cdf (RFILTER_RESOLUTION = 31) */
    /* thin lens (cameras/ThinlensCamera.cpp:85-126); the aperture texture is only ever sampled by the forward path tracer
     * (samplePosition, :85-97): the default disk (textures/DiskTexture.cpp:78-86), an n-blade polygon (BladeTexture.cpp:21-31, 110-130) or a
     * bitmap (BitmapTexture::sample) */
    int32_t type;             /* TGHIP_CAMERA_*                                   */
    float   focus_dist, aperture_size, cat_eye;
    float   inv_xf[12];       /* rows 0..2 of Camera::_invTransform (3x4, row-major, translation in column 3) */
    int32_t medium;           /* Camera::_medium (Camera.cpp:49-50): index into media[], -1 = none */
    int32_t aperture_type;    /* TGHIP_APERTURE_*                                  */
    int32_t blade_count;      /* BladeTexture::_numBlades                          */
    float   blade_angle, blade_step;   /* _angle, _bladeAngle = 2 pi / blades     */
    float   blade_edge[2];    /* _baseEdge                                         */
    /* TGHIP_APERTURE_BITMAP: the aperture bitmap's Distribution2D (ThinlensCamera::precompute makes it samplable with MAP_UNIFORM,
     * cameras/ThinlensCamera.cpp:27-35): its size and the float offset of its tables in dist[], laid out like TgHipTexture::dist_offset's */
    int32_t  aperture_w, aperture_h;
    uint32_t aperture_dist;
} TgHipCamera;

/* ---- integrator settings (TraceSettings.hpp:23-39, PathTracerSettings.hpp:25-43) -------- */
typedef struct TgHipSettings {
    int32_t min_bounces, max_bounces;
    int32_t enable_light_sampling, enable_two_sided_shading, enable_consistency_checks;
    int32_t enable_volume_light_sampling;   /* PathTracerSettings.hpp:13,38; low_order_scattering / include_surfaces must keep their defaults (true) */
    int32_t pad[2];
} TgHipSettings;

typedef struct TgHipSceneDesc {
    uint32_t abi_version;     /* TGHIP_ABI_VERSION */
    uint32_t num_nodes, num_recs, num_objects, num_lights, num_infinite_lights, num_bsdfs, num_textures;
    const TgHipBvhNode *nodes;
    const TgHipPrimRec *recs;
    const TgHipTriAttr *tri_attrs;        /* num_recs entries (unused for non-triangles) */
    const TgHipObject  *objects;
    const int32_t      *lights;           /* object indices, TraceableScene::_lights order (TraceableScene.hpp:86-102) */
    const int32_t      *infinite_lights;  /* object indices, TraceableScene::_infiniteLights */
    const TgHipBsdf    *bsdfs;
    const TgHipTexture *textures;
    const float        *texels;  uint64_t num_texel_floats;
    const float        *dist;    uint64_t num_dist_floats;
    /* sampled mesh emitters (TriangleMesh.cpp:395-409): per mesh a block of floats at objects[].first_light_tri:
     * cdf[n + 1] of the triangle areas (Distribution1D: normalised, last = 1), then n x 9 floats p0,p1,p2 (world space,
     * the mesh's own triangle order); objects[].area is the total area */
    const float        *light_tris;  uint64_t num_light_tri_floats;
    /* generator matrices of the Sobol' sequence, TGHIP_SOBOL_DIMS x TGHIP_SOBOL_BITS words -- the host's own copy of
     * sobol::Matrices::matrices (thirdparty/sobol/sobol.h:30-35); NULL/0 unless passes use TGHIP_PASS_SOBOL */
    const uint32_t     *sobol_matrices;  uint64_t num_sobol_words;
    uint32_t num_instances;               /* instance records among recs (0: single-level scene) */
    uint32_t num_top_recs;                /* top-level records = recs[0, num_top_recs): the non-instance and the instance records in the
                                             wide BVH's order, then the instance-set records; the rest belong to masters */
    /* ABI 8, scenes with instances: the leaf slots of the reference's instance trees (see the instance-set record) -> instance record,
     * and, at a leaf's FIRST slot, the leaf's box as its parent stores it: lo[3], pad, hi[3], pad -- two 16-byte loads; the root's bounds
     * for a tree that is one leaf; num_inst_prims entries, the second slot of a two-instance leaf unused */
    const uint32_t     *inst_prims;       uint32_t num_inst_prims;
    const float        *inst_leaf_boxes;
    /* ... and per top-level record that is an instance (indexed by the record: 8 floats at 8 x record index, num_top_recs entries, the
     * others zero) a box -- lo[3], pad, hi[3], pad -- that holds the instance's geometry in world space and lies inside the reference's
     * box of the instance: a ray that misses it cannot hit the instance, so its master is not walked (an optimisation only: the
     * reference walks it and finds nothing).  The wide BVH is built from these boxes. */
    const float        *inst_tight_boxes;
    const TgHipMedium  *media;  uint32_t num_media;   /* Scene::_media; NULL/0 = the scene has no participating media */
    /* the wide BVH: wide_nodes[0] is the root of the tree over recs[0, num_top_recs); with instances every master's wide
     * subtree follows (its root in the instance records' c[2]).  NULL/0 = the device walks the BVH2 (flat-list scenes do) */
    const TgHipWideNode *wide_nodes;  uint32_t num_wide_nodes;
    /* the reference's top-level Embree tree over the records of a flat list of analytic primitives (TgHipTopNode; ABI 10); NULL / 0 otherwise */
    const TgHipTopNode *top_nodes;  uint32_t num_top_nodes;
    TgHipCamera   camera;
    TgHipSettings settings;
    float         bounds_lo[3], bounds_hi[3];
} TgHipSceneDesc;

/* ---- passes ------------------------------------------------------------------------------
 * A pass renders samples [spp_begin, spp_end) of every pixel owned by this shard.  Ownership
 * follows the reference's 16x16 tile dicing (PathTraceIntegrator.cpp:27-42): tile (tx, ty) belongs to
 * shard tghip_tile_owner(tx, ty, shard_count) = (tx + ty*skew) % shard_count -- a diagonal interleave, so
 * that a shard's tiles are spread over the image in x AND y whatever the image width (plain t % N over
 * row-major tiles degenerates into N vertical stripes whenever the tiles per row are a multiple of N:
 * 1280, 1920 and 3840 pixels all are for N = 8); a shard renders its tiles in row-major order.  Random
 * numbers are a counter-based PCG stream keyed by (seed, pixelIndex, sampleIndex) -- see DESIGN.md "RNG". */
static inline uint32_t tghip_shard_skew(uint32_t shard_count)      /* the smallest odd prime that does not divide shard_count */
{
    static const uint32_t primes[6] = {3u, 5u, 7u, 11u, 13u, 17u};
    for (int i = 0; i < 6; ++i)
        if (shard_count % primes[i] != 0u)
            return primes[i];
    return 1u;
}
static inline uint32_t tghip_tile_owner(uint32_t tx, uint32_t ty, uint32_t shard_count)
{
    return shard_count <= 1u ? 0u : (tx + ty*tghip_shard_skew(shard_count)) % shard_count;
}
#define TGHIP_SOBOL_DIMS 1024u
#define TGHIP_SOBOL_BITS 52u
#define TGHIP_TILE_SIZE 16u              /* PathTraceIntegrator::TileSize */
#define TGHIP_VARIANCE_TILE_SIZE 4u      /* PathTraceIntegrator::VarianceTileSize: one SampleRecord per 4x4 pixels */
/* pass flags */
#define TGHIP_PASS_SOBOL   1u   /* next1D/next2D come from the scrambled Sobol' sequence exactly as SobolPathSampler does
                                   (sampling/SobolPathSampler.hpp:20-79: scramble = tile seed ^ hash32(pixel), permuted index,
                                   one dimension per draw up to 1024); booleans and later dimensions come from the
                                   counter-based PCG stream (the reference uses a sequential per-tile stream there) */
#define TGHIP_PASS_RECORDS 2u   /* keep the per-4x4-pixel SampleRecords (path_tracer/SampleRecord.hpp:46-65) up to date */

#define TGHIP_PASS_AUX     4u   /* keep the auxiliary output buffers (TgHipAuxPixel) up to date: depth / normal / albedo / visibility of the
                                   first non-specular vertex (PathTracer.cpp:78-96, 133-140) and the colour, each with its A/B halves and
                                   sample variance (cameras/OutputBuffer.hpp:90-132) */

#define TGHIP_PASS_SAMPLES 8u   /* parity instrumentation: additionally keep the radiance of every individual sample of the pass -- the
                                   return value of PathTracer::traceSample (PathTracer.cpp:14-149) for (pixel, sample index) -- in a
                                   device buffer read back with tghip_download_samples; not with record_index / record_count */

/* auxiliary outputs (cameras/OutputBufferSettings.cpp:8-14) and their channels in TgHipAuxPixel */
enum { TGHIP_AUX_COLOR = 0, TGHIP_AUX_DEPTH = 1, TGHIP_AUX_NORMAL = 2, TGHIP_AUX_ALBEDO = 3, TGHIP_AUX_VISIBILITY = 4, TGHIP_AUX_OUTPUTS = 5 };
#define TGHIP_AUX_CHANNELS 11u   /* color rgb 0-2 | depth 3 | normal xyz 4-6 | albedo rgb 7-9 | visibility 10 */
/* OutputBuffer state of one pixel for all five outputs, kept as if two_buffer_variance and sample_variance were both on
 * (OutputBuffer.hpp:90-132: _bufferA = running mean of the samples with even index, _bufferB = odd, _variance = Welford sum
 * against the mean of both, _sampleCount per output -- an output skips the samples that did not record it).  The host
 * derives what a scene's OutputBufferSettings ask for: mean = (A nA + B nB)/n. */
typedef struct TgHipAuxPixel {
    float    a[11], b[11], variance[11];
    uint32_t count[5];
} TgHipAuxPixel;              /* 152 B */

/* the device-resident part of SampleRecord: Welford mean / running variance of the sample luminance,
 * accumulated in the reference's order (tile row-major pixel order, then sample index: PathTraceIntegrator.cpp:136-156) */
typedef struct TgHipSampleRecord { uint32_t sample_count; float mean, running_variance; } TgHipSampleRecord;

typedef struct TgHipPassDesc {
    uint32_t spp_begin, spp_end;
    uint32_t seed;
    uint32_t shard_index, shard_count;   /* 0,1 = whole image */
    uint32_t flags;                      /* TGHIP_PASS_* */
    /* host arrays, borrowed until tghip_wait returns; NULL when unused */
    const uint32_t *tile_seeds;          /* TGHIP_PASS_SOBOL: SobolPathSampler seed per 16x16 tile, dice order */
    const uint32_t *record_index;        /* adaptive sampling: per SampleRecord (ceil(W/4) x ceil(H/4), row-major) the first */
    const uint32_t *record_count;        /*   sample index and the samples per pixel of this pass; then spp_begin/spp_end are ignored */
} TgHipPassDesc;

typedef struct TgHipCounters {
    uint64_t samples;           /* camera paths completed                         */
    uint64_t closest_rays;      /* closest-hit queries (extension rays)           */
    uint64_t shadow_rays;       /* shadow-type closest-hit queries                */
    uint64_t nodes_visited;     /* BVH nodes fetched, all rays (0 unless counting is enabled) */
    uint64_t prims_tested;      /* primitive records tested, all rays                          */
    uint64_t iterations;        /* wavefront iterations                           */
    double   ms_trace_closest, ms_trace_shadow, ms_shade, ms_other, ms_total;  /* HIP-event time, last pass set */
    uint64_t launches_trace_closest, launches_trace_shadow, launches_shade;
    uint64_t nodes_visited_shadow; /* the shadow-ray share of nodes_visited / prims_tested (counting enabled) */
    uint64_t prims_tested_shadow;
    uint64_t shadow_slots;         /* path vertices that queued at least one shadow ray */
    uint64_t tail_launches;        /* batches whose last paths ran in the single-launch tail kernel (k_tail) */
} TgHipCounters;

/* closest-hit query record for tghip_trace_rays (and the oracle's equivalent) */
typedef struct TgHipRay { float o[3], tmin, d[3], tmax; } TgHipRay;          /* 32 B */
typedef struct TgHipHit { float t, u, v; int32_t rec; } TgHipHit;            /* 16 B; rec = -1 on miss */

typedef struct tghip_ctx tghip_ctx;

tghip_ctx  *tghip_create(int device_ordinal);
void        tghip_destroy(tghip_ctx *ctx);
const char *tghip_last_error(tghip_ctx *ctx);      /* ctx may be NULL: last create error */
int         tghip_device_count(void);

int tghip_upload_scene(tghip_ctx *ctx, const TgHipSceneDesc *scene);
int tghip_render_pass(tghip_ctx *ctx, const TgHipPassDesc *pass);   /* asynchronous */
int tghip_wait(tghip_ctx *ctx);
int tghip_abort(tghip_ctx *ctx);
int tghip_clear_framebuffer(tghip_ctx *ctx);
/* Use caller-owned DEVICE buffers (e.g. a torch tensor) as framebuffer: sum = W*H*3 floats,
 * count = W*H uint32.  Pass NULLs to go back to the internal buffers. */
int tghip_bind_framebuffer(tghip_ctx *ctx, float *dev_rgb_sum, uint32_t *dev_count);
int tghip_download_framebuffer(tghip_ctx *ctx, float *rgb_sum, uint32_t *count, size_t npixels);
/* restores a saved framebuffer (OutputBuffer::deserialize, cameras/OutputBuffer.hpp:191-200): resume of an interrupted render */
int tghip_upload_framebuffer(tghip_ctx *ctx, const float *rgb_sum, const uint32_t *count, size_t npixels);
/* SampleRecords (TGHIP_PASS_RECORDS): n = ceil(W/4)*ceil(H/4); cleared by tghip_clear_framebuffer.  Records of tiles
 * this context never rendered stay zero; upload restores a resumed / merged state. */
int tghip_download_records(tghip_ctx *ctx, TgHipSampleRecord *out, size_t n);
int tghip_upload_records(tghip_ctx *ctx, const TgHipSampleRecord *in, size_t n);
/* auxiliary output buffers (TGHIP_PASS_AUX), one TgHipAuxPixel per image pixel; allocated by the first TGHIP_PASS_AUX pass,
 * cleared by tghip_clear_framebuffer (Camera::serializeOutputBuffers / deserializeOutputBuffers, Camera.cpp:222-238) */
int tghip_download_aux(tghip_ctx *ctx, TgHipAuxPixel *out, size_t npixels);
int tghip_upload_aux(tghip_ctx *ctx, const TgHipAuxPixel *in, size_t npixels);
/* the per-sample radiance of the last TGHIP_PASS_SAMPLES pass: nfloats = W*H*(spp_end - spp_begin)*3, laid out
 * [pixel (row-major)][sample - spp_begin][rgb]; samples of pixels the pass's shard does not own are zero */
int tghip_download_samples(tghip_ctx *ctx, float *rgb, size_t nfloats);
/* Multi-GPU framebuffer merge inside one process: ctxs[0..n) are the contexts (one per device, all with the same scene
 * uploaded) that rendered the tile shards 0..n-1 of a frame (TgHipPassDesc.shard_index/shard_count).  Their radiance sums
 * (float32) and sample counts (uint32) are sum-reduced by RCCL (ncclReduce over xGMI, one communicator per device, created at
 * the first call and kept) into a scratch buffer on ctxs[root]'s device and copied from there into rgb_sum / count (host,
 * npixels = W*H; either may be NULL).  Tile ownership is disjoint, so the sums are exact (x + 0) in any reduction order; the
 * contexts' own framebuffers are left as they are, so passes can go on accumulating.  librccl.so is loaded at the first call;
 * TGHIP_E_UNSUPPORTED when it is missing. */
int tghip_reduce_framebuffers(tghip_ctx *const *ctxs, int n, int root, float *rgb_sum, uint32_t *count, size_t npixels);
/* The same merge between PROCESSES, one per GPU (how `python -m torch.distributed.run --nproc-per-node N` / mpirun deploy it; the reference's
 * analogue is one `tungsten` process per machine and `hdrmanip --merge` over their output files, src/hdrmanip/hdrmanip.cpp:69-112):
 *   rank 0:     tghip_comm_unique_id(id, TGHIP_COMM_ID_BYTES)  -- ncclGetUniqueId; the caller carries the bytes to the other ranks by any means it
 *               has (a file, a socket, MPI_Bcast, torch.distributed.broadcast_object_list);
 *   every rank: tghip_comm_init_rank(ctx, id, bytes, nranks, rank)  -- collective: ncclCommInitRank on ctx's device; the communicator belongs to
 *               the context and is destroyed with it;
 *   every rank: tghip_reduce_framebuffer_rank(ctx, root, rgb_sum, count, npixels)  -- collective: waits for ctx's pass, then ncclReduce (sum) of the
 *               radiance sums and sample counts of every rank's framebuffer -- rank r rendered shard r of `nranks`, TgHipPassDesc.shard_* -- into a
 *               scratch image on the root's device; on the root that image is copied to rgb_sum / count (host; either may be NULL: the reduced
 *               image then stays in HBM), the other ranks pass NULL.  Exact for the same reason as above; framebuffers are left as they are. */
#define TGHIP_COMM_ID_BYTES 128
int tghip_comm_unique_id(void *id, size_t bytes);
int tghip_comm_init_rank(tghip_ctx *ctx, const void *id, size_t bytes, int nranks, int rank);
int tghip_reduce_framebuffer_rank(tghip_ctx *ctx, int root, float *rgb_sum, uint32_t *count, size_t npixels);
int tghip_trace_rays(tghip_ctx *ctx, const TgHipRay *rays, TgHipHit *hits, size_t n, int repeats, double *ms_per_launch);
/* Self-test of the device's libm restatements (csrc/hip/pt_libm.h: glibc's sinf / cosf / logf / expf / atan2f / powf / cbrtf as the shading
 * kernels call them, pt_math.h: acosf): y[i] = fn(x[i]) evaluated ON THE DEVICE, host pointers.  fn: TGHIP_LIBM_*.  The two-argument
 * functions read their operands interleaved from x (2 n floats): ATAN2F y[i] = atan2f(x[2i], x[2i+1]), POWF y[i] = powf(x[2i], x[2i+1]).
 * EMBREE_RCP / RCPPS: Embree's rcp(a) = r*(2 - r*a) and its r = Intel's RCPPS estimate as the triangle test restates them (pt_scene.h: embreeRcp,
 * rcppsIntel).  The GPU tests compare y with the host libm / the oracle's restatement bit for bit. */
enum { TGHIP_LIBM_SINF = 0, TGHIP_LIBM_COSF = 1, TGHIP_LIBM_LOGF = 2, TGHIP_LIBM_EXPF = 3, TGHIP_LIBM_SINCOS_SIN = 4, TGHIP_LIBM_SINCOS_COS = 5,
       TGHIP_LIBM_ACOSF = 6, TGHIP_LIBM_ATAN2F = 7, TGHIP_LIBM_POWF = 8, TGHIP_LIBM_CBRTF = 9,
       TGHIP_LIBM_EMBREE_RCP = 10, TGHIP_LIBM_RCPPS = 11, TGHIP_LIBM_TANF = 12,
       /* double precision (AtmosphericMedium::inverseOpticalDepth: std::exp / log / erf / sqrt on doubles): x and y then point to n DOUBLES */
       TGHIP_LIBM_EXPD = 13, TGHIP_LIBM_LOGD = 14, TGHIP_LIBM_ERFD = 15, TGHIP_LIBM_SQRTD = 16 };
int tghip_debug_libm(tghip_ctx *ctx, int fn, const float *x, float *y, size_t n);
int tghip_set_option(tghip_ctx *ctx, const char *key, long long value);  /* "count_traversal", "max_slots", ...; "top_tree" = 0 before an upload:
                                                                             TgHipSceneDesc::top_nodes is ignored, flat lists are walked in record order.
                                                                             Instrumentation that never changes an image (tests hold that): "lds_tables" = 0
                                                                             shades as if the small tables did not fit the LDS copy, "env_lds" = 0 samples the
                                                                             environment map through its global tables, "hoist_quad" = 0 leaves a scene's one
                                                                             quad inside the walks, "tail_family" = 0 runs the all-types tail kernel.
                                                                             Round 6, instanced scenes: "inst_wide" = 0 walks masters through their BVH2 (round 5's
                                                                             kernel; images differ only where two triangles of a master answer a ray one rounding
                                                                             apart), "inst_phase_min" / "inst_refill_at" the phase vote's threshold and the refill
                                                                             level of k_trace_closest_instw, "inst_shadow_fast" = 0 shadow rays on
                                                                             k_trace_shadow_wide<., ., INST>; "shade_lds_pad" bytes of unused LDS per shading
                                                                             workgroup (an occupancy throttle for experiments) */
int tghip_get_counters(tghip_ctx *ctx, TgHipCounters *out);
int tghip_reset_counters(tghip_ctx *ctx);
/* Instrumentation (no reference analogue; bench.py roofline.valu.walk, profiles/r6_lane_util.json): the tallies the counting variants of the
 * two wide walks keep ("count_traversal" = 1), summed over every wave launch since the last tghip_reset_counters.  walk 0 = closest hit, 1 = shadow.
 * out[0..n): [0..3] wave time in 10-ns ticks (queue expansion, loop with queue, loop after the queue ran dry, wait + write-back), [4] wave launches,
 * [5] / [6] loop turns before / after dry, [7] / [8] busy lanes summed over those turns, [9] / [10] walks suspended / resumed, [11] longest loop,
 * [12] / [13] turns that ran the record test / lanes with a record in them, [14] / [15] the same for the node visit, [16] / [17] refill blocks run /
 * lanes refilled, [18] / [19] publish (NEE-term) blocks run / lanes in them, [20] record tests accepted, [21] rays walked.  Instanced scenes (walk 0):
 * [22] wave launches of k_trace_closest_instw (0: k_trace_closest_inst ran), [23] wide nodes visited inside masters, [24 + 2 k] / [25 + 2 k] runs / lanes
 * of section k of the turn (pt_wavefront.h lists the twelve sections of either kernel; bench.py: walk_summary names them).  Returns the number of
 * values written (<= n), or a negative error. */
int tghip_get_walk_stats(tghip_ctx *ctx, int walk, uint64_t *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* TUNGSTEN_HIP_H_ */
