"""ctypes access to oracle/liboracle.so -- TEST INFRASTRUCTURE (checker only, never the thing measured)."""
import ctypes as C
import os

import numpy as np

from tungsten_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


class OracleCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64)]


DESC_P = C.POINTER(capi.TgHipSceneDesc)
_lib.oracle_trace_sample.argtypes = [DESC_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
_lib.oracle_trace_sample_sobol.argtypes = [DESC_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
_lib.oracle_render.argtypes = [DESC_P, C.POINTER(capi.TgHipPassDesc), C.c_void_p, C.c_void_p, C.POINTER(OracleCounters), C.c_int]
_lib.oracle_render.restype = C.c_int
_lib.oracle_trace_rays.argtypes = [DESC_P, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
_lib.oracle_rng_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
_lib.oracle_camera_ray.argtypes = [DESC_P, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
_lib.oracle_bsdf_eval.argtypes = [DESC_P, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_float)]
_lib.oracle_bsdf_sample.argtypes = [DESC_P, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
_lib.oracle_bsdf_sample.restype = C.c_int
_lib.oracle_light_sample.argtypes = [DESC_P, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
_lib.oracle_light_sample.restype = C.c_int
_lib.oracle_texture_eval.argtypes = [DESC_P, C.c_int, C.c_float, C.c_float, C.c_void_p]


RECORD_DTYPE = np.dtype([("sample_count", np.uint32), ("next_sample_count", np.uint32), ("sample_index", np.uint32),
                         ("adaptive_weight", np.float32), ("mean", np.float32), ("running_variance", np.float32)])
DEVICE_RECORD_DTYPE = np.dtype([("sample_count", np.uint32), ("mean", np.float32), ("running_variance", np.float32)])
_lib.oracle_render_records.argtypes = [DESC_P, C.POINTER(capi.TgHipPassDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OracleCounters), C.c_int]
_lib.oracle_render_records.restype = C.c_int
_lib.oracle_generate_work.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_int]
_lib.oracle_generate_work.restype = C.c_int
_lib.oracle_dice_tiles.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_void_p]
_lib.oracle_dice_tiles.restype = C.c_uint64
_lib.oracle_integrate.argtypes = [DESC_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int]
_lib.oracle_integrate.restype = C.c_int
_lib.oracle_render_aux.argtypes = [DESC_P, C.POINTER(capi.TgHipPassDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OracleCounters), C.c_int]
_lib.oracle_render_aux.restype = C.c_int
_lib.oracle_integrate_aux.argtypes = [DESC_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int]
_lib.oracle_integrate_aux.restype = C.c_int


def dice_tiles(width, height, seed):
    """(tile seeds, state of the integrator's sampler afterwards)."""
    seeds = np.zeros(((width + 15)//16)*((height + 15)//16), np.uint32)
    state = _lib.oracle_dice_tiles(width, height, seed & 0xFFFFFFFF, seeds.ctypes.data)
    return seeds, int(state)


def generate_work(records, width, height, sampler_state, current_spp, next_spp, adaptive):
    """In-place PathTraceIntegrator::generateWork on a RECORD_DTYPE array; returns (has work, new sampler state)."""
    assert records.dtype == RECORD_DTYPE and records.flags.c_contiguous
    st = C.c_uint64(sampler_state)
    r = _lib.oracle_generate_work(records.ctypes.data, width, height, C.byref(st), current_spp, next_spp, int(bool(adaptive)))
    return bool(r), int(st.value)


def render_pass(desc, width, height, seed, spp_begin=0, spp_end=0, flags=0, tile_seeds=None, record_index=None, record_count=None,
                records=None, ssum=None, count=None, shard_index=0, shard_count=1, threads=0, counters=None):
    """oracle_render_records with the full TgHipPassDesc; accumulates into ssum/count/records when given."""
    ssum = np.zeros((height, width, 3), np.float32) if ssum is None else ssum
    count = np.zeros((height, width), np.uint32) if count is None else count
    p = capi.TgHipPassDesc(spp_begin, spp_end, seed & 0xFFFFFFFF, shard_index, shard_count, flags)
    keep = []
    for name, arr in (("tile_seeds", tile_seeds), ("record_index", record_index), ("record_count", record_count)):
        if arr is not None:
            a = np.ascontiguousarray(arr, np.uint32)
            keep.append(a)
            setattr(p, name, a.ctypes.data_as(C.POINTER(C.c_uint32)))
    rc = _lib.oracle_render_records(desc, C.byref(p), ssum.ctypes.data, count.ctypes.data,
                                    records.ctypes.data if records is not None else None,
                                    C.byref(counters) if counters is not None else None, threads)
    if rc != 0:
        raise RuntimeError("oracle_render_records failed (%d)" % rc)
    return ssum, count


def integrate(desc, width, height, seed, spp, spp_step, adaptive, sobol, threads=0):
    """The whole CLI render loop; returns (sum, count, records[passes, vh, vw], spp after each pass)."""
    vw, vh = (width + 3)//4, (height + 3)//4
    max_passes = (spp + spp_step - 1)//spp_step
    ssum = np.zeros((height, width, 3), np.float32)
    count = np.zeros((height, width), np.uint32)
    rec = np.zeros((max_passes, vh, vw), RECORD_DTYPE)
    pass_spp = np.zeros(max_passes, np.uint32)
    n = _lib.oracle_integrate(desc, seed & 0xFFFFFFFF, spp, spp_step, int(bool(adaptive)), int(bool(sobol)), ssum.ctypes.data, count.ctypes.data,
                              rec.ctypes.data, max_passes, pass_spp.ctypes.data, threads)
    if n < 0:
        raise RuntimeError("oracle_integrate failed (%d)" % n)
    return ssum, count, rec[:n], pass_spp[:n]


AUX_DTYPE = np.dtype([("a", np.float32, 11), ("b", np.float32, 11), ("variance", np.float32, 11), ("count", np.uint32, 5)])


def integrate_aux(desc, width, height, seed, spp, spp_step, adaptive, sobol, threads=0):
    """integrate() for a scene with renderer.output_buffers: additionally returns the output buffers, [H, W] of AUX_DTYPE
    (include/tungsten_hip.h: TgHipAuxPixel)."""
    vw, vh = (width + 3)//4, (height + 3)//4
    max_passes = (spp + spp_step - 1)//spp_step
    ssum = np.zeros((height, width, 3), np.float32)
    count = np.zeros((height, width), np.uint32)
    rec = np.zeros((max_passes, vh, vw), RECORD_DTYPE)
    pass_spp = np.zeros(max_passes, np.uint32)
    aux = np.zeros((height, width), AUX_DTYPE)
    n = _lib.oracle_integrate_aux(desc, C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32(spp), C.c_uint32(spp_step), int(bool(adaptive)), int(bool(sobol)),
                                  C.c_void_p(ssum.ctypes.data), C.c_void_p(count.ctypes.data), C.c_void_p(rec.ctypes.data), max_passes,
                                  C.c_void_p(pass_spp.ctypes.data), C.c_void_p(aux.ctypes.data), threads)
    if n < 0:
        raise RuntimeError("oracle_integrate_aux failed (%d)" % n)
    return ssum, count, rec[:n], pass_spp[:n], aux


def render(desc, width, height, spp_begin, spp_end, seed, shard_index=0, shard_count=1, threads=0, counters=None):
    """Returns (sum[H,W,3], count[H,W]) like the device framebuffer."""
    ssum = np.zeros((height, width, 3), np.float32)
    count = np.zeros((height, width), np.uint32)
    p = capi.TgHipPassDesc(spp_begin, spp_end, seed & 0xFFFFFFFF, shard_index, shard_count, 0)
    c = counters if counters is not None else OracleCounters()
    _lib.oracle_render(desc, C.byref(p), ssum.ctypes.data, count.ctypes.data, C.byref(c), threads)
    return ssum, count


def trace_sample(desc, seed, px, py, sample, tile_seed=None):
    """One PathTracer::traceSample; tile_seed selects the Sobol' sampler of that tile (needs sobol_matrices in desc)."""
    rgb = (C.c_float*3)()
    if tile_seed is None:
        _lib.oracle_trace_sample(desc, seed & 0xFFFFFFFF, px, py, sample, rgb)
    else:
        _lib.oracle_trace_sample_sobol(desc, seed & 0xFFFFFFFF, int(tile_seed), px, py, sample, rgb)
    return np.array(rgb[:], np.float32)


def trace_rays(desc, rays, wide=False):
    """Closest hits + exact node / record visit counts.  wide: walk the scene's 8-wide BVH (when it has one) with the
    device's per-ray machine instead of the BVH2."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    hits = np.empty(rays.shape[0], dtype=[("t", np.float32), ("u", np.float32), ("v", np.float32), ("rec", np.int32)])
    nodes, prims = C.c_uint64(0), C.c_uint64(0)
    _lib.oracle_set_wide_bvh(1 if wide else 0)
    try:
        _lib.oracle_trace_rays(desc, rays.ctypes.data, hits.ctypes.data, rays.shape[0], C.byref(nodes), C.byref(prims))
    finally:
        _lib.oracle_set_wide_bvh(0)
    return hits, nodes.value, prims.value


def rng_stream(seed, pixel, sample, n):
    out = np.empty(n, np.float32)
    _lib.oracle_rng_stream(seed & 0xFFFFFFFF, pixel, sample, n, out.ctypes.data)
    return out


def camera_ray(desc, px, py, xi0, xi1):
    o, d = np.empty(3, np.float32), np.empty(3, np.float32)
    _lib.oracle_camera_ray(desc, px, py, xi0, xi1, o.ctypes.data, d.ctypes.data)
    return o, d


def bsdf_eval(desc, bsdf, wi, wo, uv, requested):
    wi, wo, uv = [np.ascontiguousarray(v, np.float32) for v in (wi, wo, uv)]
    f = np.empty(3, np.float32)
    pdf = C.c_float(0)
    _lib.oracle_bsdf_eval(desc, bsdf, wi.ctypes.data, wo.ctypes.data, uv.ctypes.data, requested & 0xFFFFFFFF, f.ctypes.data, C.byref(pdf))
    return f, pdf.value


def bsdf_sample(desc, bsdf, wi, uv, requested, xi):
    wi, uv, xi = [np.ascontiguousarray(v, np.float32) for v in (wi, uv, xi)]
    wo, weight = np.empty(3, np.float32), np.empty(3, np.float32)
    pdf, lobe, consumed = C.c_float(0), C.c_uint32(0), C.c_int(0)
    ok = _lib.oracle_bsdf_sample(desc, bsdf, wi.ctypes.data, uv.ctypes.data, requested & 0xFFFFFFFF, xi.ctypes.data, len(xi),
                                 wo.ctypes.data, weight.ctypes.data, C.byref(pdf), C.byref(lobe), C.byref(consumed))
    return bool(ok), wo, weight, pdf.value, lobe.value, consumed.value


def light_sample(desc, light, p, xi0, xi1):
    p = np.ascontiguousarray(p, np.float32)
    d = np.empty(3, np.float32)
    dist, pdf = C.c_float(0), C.c_float(0)
    ok = _lib.oracle_light_sample(desc, light, p.ctypes.data, xi0, xi1, d.ctypes.data, C.byref(dist), C.byref(pdf))
    return bool(ok), d, dist.value, pdf.value


_lib.oracle_leaf_bounds.argtypes = [DESC_P, C.c_uint32, C.c_void_p, C.c_void_p]
_lib.oracle_leaf_bounds.restype = C.c_int
_lib.oracle_flat_device_form.argtypes = [DESC_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
_lib.oracle_flat_device_form.restype = C.c_size_t
_lib.oracle_set_flat_order.argtypes = [C.c_int]


def leaf_bounds(desc, rec):
    """(lo, hi) of the record's leaf box in the reference's Embree BVH, or None for a kind whose bounds are not restated."""
    lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
    return (lo, hi) if _lib.oracle_leaf_bounds(desc, rec, lo.ctypes.data, hi.ctypes.data) else None


def flat_device_form(desc, rays):
    """The device's formulation of the flat-list walk (oracle.c: oracle_flat_device_form): (hits, decided, rays on which it differs from the walk)."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    hits = np.empty(rays.shape[0], dtype=[("t", np.float32), ("u", np.float32), ("v", np.float32), ("rec", np.int32)])
    decided = np.zeros(rays.shape[0], np.uint8)
    differing = _lib.oracle_flat_device_form(desc, rays.ctypes.data, hits.ctypes.data, decided.ctypes.data, rays.shape[0])
    return hits, decided.astype(bool), int(differing)


def trace_rays_plain_list(desc, rays):
    """Closest hits of the plain list in record order (what flat lists were walked as before the visiting order of Embree's leaves was restated)."""
    _lib.oracle_set_flat_order(0)
    try:
        return trace_rays(desc, rays)[0]
    finally:
        _lib.oracle_set_flat_order(1)


_lib.oracle_flat_device_form2.argtypes = [DESC_P, C.c_void_p, C.c_void_p, C.c_size_t]
_lib.oracle_flat_device_form2.restype = C.c_size_t


def flat_device_form2(desc, rays):
    """The cheaper shortcut sketched for the next round (oracle.c: flat_shortcut_decides_v2; no device counterpart yet): (decided, rays on which a
    decided answer differs from the walk)."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    decided = np.zeros(rays.shape[0], np.uint8)
    differing = _lib.oracle_flat_device_form2(desc, rays.ctypes.data, decided.ctypes.data, rays.shape[0])
    return decided.astype(bool), int(differing)
