"""ctypes mirror of include/tungsten_hip.h and include/tungsten_host.h.

This module only *describes* the C-ABI (struct layouts, prototypes) and loads the in-tree shared
library built by ``__graft_entry__.build()`` / ``make``.  There is no Python or CPU fallback for
any compute entry point: if the library is missing, importing :mod:`tungsten_amd` fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtungsten_hip.so")

f32, i32, u32, i64, u64 = C.c_float, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64


class TgHipBvhNode(C.Structure):
    _fields_ = [("lo0", f32*3), ("hi0", f32*3), ("lo1", f32*3), ("hi1", f32*3),
                ("child0", i32), ("child1", i32), ("pad", u32*2)]


class TgHipWideNode(C.Structure):
    _fields_ = [("origin", f32*3), ("exp", C.c_uint8*3), ("imask", C.c_uint8), ("child_base", u32), ("rec_base", u32),
                ("leaf_valid", u32), ("reserved", u32), ("qlo", (C.c_uint8*8)*3), ("qhi", (C.c_uint8*8)*3)]


class TgHipPrimRec(C.Structure):
    _fields_ = [("a", f32*3), ("meta", u32), ("b", f32*3), ("p0", f32), ("c", f32*3), ("p1", f32)]


class TgHipTriAttr(C.Structure):
    _fields_ = [("n0", f32*3), ("n1", f32*3), ("n2", f32*3), ("uv0", f32*2), ("uv1", f32*2), ("uv2", f32*2),
                ("bsdf", i32)]


class TgHipObject(C.Structure):
    _fields_ = [("type", i32), ("bsdf", i32), ("emission", i32), ("light", i32), ("flags", u32),
                ("area", f32), ("inv_area", f32), ("first_light_tri", i32),
                ("base", f32*3), ("edge0", f32*3), ("edge1", f32*3), ("normal", f32*3), ("inv_uv_sq", f32*2),
                ("pos", f32*3), ("scale", f32*3), ("rot", f32*9), ("face_cdf", f32*3), ("num_light_tris", i32), ("int_medium", i32), ("ext_medium", i32)]


class TgHipBsdf(C.Structure):
    _fields_ = [("type", i32), ("lobes", u32), ("albedo", i32), ("distribution", i32), ("roughness", i32),
                ("sub0", i32), ("sub1", i32), ("tex1", i32),
                ("ior", f32), ("thickness", f32), ("avg_transmittance", f32), ("diffuse_fresnel", f32),
                ("enable_refraction", i32), ("eta", f32*3), ("k", f32*3), ("sigma_a", f32*3),
                ("scaled_sigma_a", f32*3), ("bump1", i32), ("pad", f32)]


class TgHipTexture(C.Structure):
    _fields_ = [("type", i32), ("flags", u32), ("w", i32), ("h", i32), ("value", f32*3), ("scale", f32),
                ("on_color", f32*3), ("res_u", i32), ("off_color", f32*3), ("res_v", i32),
                ("avg", f32*3), ("pad", f32), ("texel_offset", i64), ("dist_offset", i64)]


class TgHipMedium(C.Structure):
    _fields_ = [("sigma_a", f32*3), ("sigma_s", f32*3), ("sigma_t", f32*3), ("absorption_only", i32), ("max_bounce", i32),
                ("phase_type", i32), ("phase_g", f32), ("trans_type", i32), ("trans_p", f32*3),
                ("medium_type", i32), ("falloff_scale", f32), ("unit_point", f32*3), ("falloff_dir", f32*3), ("pad", f32*3)]


class TgHipCamera(C.Structure):
    _fields_ = [("pos", f32*3), ("plane_dist", f32), ("xf", f32*9), ("ratio", f32), ("pixel_size_x", f32),
                ("res_x", i32), ("res_y", i32), ("filter_type", i32), ("filter_width", f32),
                ("filter_bin_size", f32), ("filter_cdf", f32*32),
                ("type", i32), ("focus_dist", f32), ("aperture_size", f32), ("cat_eye", f32), ("inv_xf", f32*12), ("medium", i32),
                ("aperture_type", i32), ("blade_count", i32), ("blade_angle", f32), ("blade_step", f32), ("blade_edge", f32*2),
                ("aperture_w", i32), ("aperture_h", i32), ("aperture_dist", u32)]


class TgHipSettings(C.Structure):
    _fields_ = [("min_bounces", i32), ("max_bounces", i32), ("enable_light_sampling", i32),
                ("enable_two_sided_shading", i32), ("enable_consistency_checks", i32), ("enable_volume_light_sampling", i32), ("pad", i32*2)]


class TgHipTopNode(C.Structure):
    _fields_ = [("lower", (f32*3)*4), ("upper", (f32*3)*4), ("child", i32*4)]


TGHIP_TOP_EMPTY = 0x7FFFFFFF


class TgHipSceneDesc(C.Structure):
    _fields_ = [("abi_version", u32), ("num_nodes", u32), ("num_recs", u32), ("num_objects", u32),
                ("num_lights", u32), ("num_infinite_lights", u32), ("num_bsdfs", u32), ("num_textures", u32),
                ("nodes", C.POINTER(TgHipBvhNode)), ("recs", C.POINTER(TgHipPrimRec)),
                ("tri_attrs", C.POINTER(TgHipTriAttr)), ("objects", C.POINTER(TgHipObject)),
                ("lights", C.POINTER(i32)), ("infinite_lights", C.POINTER(i32)),
                ("bsdfs", C.POINTER(TgHipBsdf)), ("textures", C.POINTER(TgHipTexture)),
                ("texels", C.POINTER(f32)), ("num_texel_floats", u64),
                ("dist", C.POINTER(f32)), ("num_dist_floats", u64),
                ("light_tris", C.POINTER(f32)), ("num_light_tri_floats", u64),
                ("sobol_matrices", C.POINTER(u32)), ("num_sobol_words", u64),
                ("num_instances", u32), ("num_top_recs", u32),
                ("inst_prims", C.POINTER(u32)), ("num_inst_prims", u32),
                ("inst_leaf_boxes", C.POINTER(f32)), ("inst_tight_boxes", C.POINTER(f32)),
                ("media", C.POINTER(TgHipMedium)), ("num_media", u32),
                ("wide_nodes", C.POINTER(TgHipWideNode)), ("num_wide_nodes", u32),
                ("top_nodes", C.POINTER(TgHipTopNode)), ("num_top_nodes", u32),
                ("camera", TgHipCamera), ("settings", TgHipSettings),
                ("bounds_lo", f32*3), ("bounds_hi", f32*3)]


TGHIP_PASS_SOBOL, TGHIP_PASS_RECORDS, TGHIP_PASS_AUX, TGHIP_PASS_SAMPLES = 1, 2, 4, 8
(TGHIP_LIBM_SINF, TGHIP_LIBM_COSF, TGHIP_LIBM_LOGF, TGHIP_LIBM_EXPF, TGHIP_LIBM_SINCOS_SIN, TGHIP_LIBM_SINCOS_COS,
 TGHIP_LIBM_ACOSF, TGHIP_LIBM_ATAN2F, TGHIP_LIBM_POWF, TGHIP_LIBM_CBRTF, TGHIP_LIBM_EMBREE_RCP, TGHIP_LIBM_RCPPS, TGHIP_LIBM_TANF,
 TGHIP_LIBM_EXPD, TGHIP_LIBM_LOGD, TGHIP_LIBM_ERFD, TGHIP_LIBM_SQRTD) = range(17)


class TgHipAuxPixel(C.Structure):
    _fields_ = [("a", f32*11), ("b", f32*11), ("variance", f32*11), ("count", u32*5)]


class TgHipPassDesc(C.Structure):
    _fields_ = [("spp_begin", u32), ("spp_end", u32), ("seed", u32), ("shard_index", u32), ("shard_count", u32),
                ("flags", u32),
                ("tile_seeds", C.POINTER(u32)), ("record_index", C.POINTER(u32)), ("record_count", C.POINTER(u32))]


class TgHipSampleRecord(C.Structure):
    _fields_ = [("sample_count", u32), ("mean", f32), ("running_variance", f32)]


class TgHostSampleRecord(C.Structure):
    _fields_ = [("sample_count", u32), ("next_sample_count", u32), ("sample_index", u32),
                ("adaptive_weight", f32), ("mean", f32), ("running_variance", f32)]


class TgHipCounters(C.Structure):
    _fields_ = [("samples", u64), ("closest_rays", u64), ("shadow_rays", u64), ("nodes_visited", u64),
                ("prims_tested", u64), ("iterations", u64),
                ("ms_trace_closest", C.c_double), ("ms_trace_shadow", C.c_double), ("ms_shade", C.c_double),
                ("ms_other", C.c_double), ("ms_total", C.c_double),
                ("launches_trace_closest", u64), ("launches_trace_shadow", u64), ("launches_shade", u64),
                ("nodes_visited_shadow", u64), ("prims_tested_shadow", u64), ("shadow_slots", u64), ("tail_launches", u64)]


class TgHipRay(C.Structure):
    _fields_ = [("o", f32*3), ("tmin", f32), ("d", f32*3), ("tmax", f32)]


class TgHipHit(C.Structure):
    _fields_ = [("t", f32), ("u", f32), ("v", f32), ("rec", i32)]


class TgHostSceneInfo(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("spp", u32), ("spp_step", u32),
                ("num_nodes", u32), ("num_recs", u32), ("num_objects", u32), ("num_lights", u32),
                ("num_bsdfs", u32), ("num_textures", u32), ("bvh_depth", i32),
                ("bvh_sah_cost", C.c_double), ("build_seconds", C.c_double),
                ("adaptive_sampling", i32), ("stratified_sampler", i32), ("current_spp", u32)]


# every symbol the two headers declare: name -> (restype, argtypes)
VP = C.c_void_p
TGHIP_COMM_ID_BYTES = 128

PROTOTYPES = {
    # include/tungsten_hip.h
    "tghip_create": (VP, [C.c_int]),
    "tghip_destroy": (None, [VP]),
    "tghip_last_error": (C.c_char_p, [VP]),
    "tghip_device_count": (C.c_int, []),
    "tghip_upload_scene": (C.c_int, [VP, C.POINTER(TgHipSceneDesc)]),
    "tghip_render_pass": (C.c_int, [VP, C.POINTER(TgHipPassDesc)]),
    "tghip_wait": (C.c_int, [VP]),
    "tghip_abort": (C.c_int, [VP]),
    "tghip_clear_framebuffer": (C.c_int, [VP]),
    "tghip_bind_framebuffer": (C.c_int, [VP, VP, VP]),
    "tghip_download_framebuffer": (C.c_int, [VP, VP, VP, C.c_size_t]),
    "tghip_upload_framebuffer": (C.c_int, [VP, VP, VP, C.c_size_t]),
    "tghip_download_records": (C.c_int, [VP, VP, C.c_size_t]),
    "tghip_upload_records": (C.c_int, [VP, VP, C.c_size_t]),
    "tghip_download_aux": (C.c_int, [VP, VP, C.c_size_t]),
    "tghip_upload_aux": (C.c_int, [VP, VP, C.c_size_t]),
    "tghip_download_samples": (C.c_int, [VP, VP, C.c_size_t]),
    "tghip_reduce_framebuffers": (C.c_int, [C.POINTER(VP), C.c_int, C.c_int, VP, VP, C.c_size_t]),
    "tghip_trace_rays": (C.c_int, [VP, VP, VP, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "tghip_debug_libm": (C.c_int, [VP, C.c_int, VP, VP, C.c_size_t]),
    "tghip_set_option": (C.c_int, [VP, C.c_char_p, C.c_longlong]),
    "tghip_get_counters": (C.c_int, [VP, C.POINTER(TgHipCounters)]),
    "tghip_reset_counters": (C.c_int, [VP]),
    "tghip_comm_unique_id": (C.c_int, [VP, C.c_size_t]),
    "tghip_comm_init_rank": (C.c_int, [VP, VP, C.c_size_t, C.c_int, C.c_int]),
    "tghip_reduce_framebuffer_rank": (C.c_int, [VP, C.c_int, VP, VP, C.c_size_t]),
    "tghip_get_walk_stats": (C.c_int, [VP, C.c_int, C.POINTER(C.c_uint64), C.c_int]),
    # include/tungsten_host.h
    "tgh_scene_load": (VP, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "tgh_scene_desc": (C.POINTER(TgHipSceneDesc), [VP]),
    "tgh_scene_info": (C.c_int, [VP, C.POINTER(TgHostSceneInfo)]),
    "tgh_scene_free": (None, [VP]),
    "tgh_renderer_open": (VP, [C.c_char_p, u32, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "tgh_renderer_context": (VP, [VP, C.c_int]),
    "tgh_renderer_info": (C.c_int, [VP, C.POINTER(TgHostSceneInfo)]),
    "tgh_renderer_step": (C.c_int, [VP, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "tgh_renderer_render": (C.c_int, [VP, C.POINTER(C.c_double), C.c_char_p, C.c_size_t]),
    "tgh_renderer_image": (C.c_int, [VP, VP, VP, VP, C.c_size_t, C.c_char_p, C.c_size_t]),
    "tgh_renderer_save_outputs": (C.c_int, [VP, C.c_char_p, C.c_size_t]),
    "tgh_renderer_close": (None, [VP]),
    "tgh_renderer_save_resume_data": (C.c_int, [VP, C.c_char_p, C.c_size_t]),
    "tgh_renderer_resume": (C.c_int, [VP, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "tgh_renderer_records": (C.c_int, [VP, VP, C.c_size_t, C.c_char_p, C.c_size_t]),
    "tgh_renderer_output_buffers": (C.c_int, [VP, VP, C.c_size_t, C.c_char_p, C.c_size_t]),
    "tgh_scheduler_create": (VP, [u32, u32, u32]),
    "tgh_scheduler_num_tiles": (C.c_size_t, [VP]),
    "tgh_scheduler_num_records": (C.c_size_t, [VP]),
    "tgh_scheduler_tile_seeds": (C.POINTER(u32), [VP]),
    "tgh_scheduler_records": (C.POINTER(TgHostSampleRecord), [VP]),
    "tgh_scheduler_generate_work": (C.c_int, [VP, u32, u32, C.c_int]),
    "tgh_scheduler_sampler_state": (u64, [VP]),
    "tgh_scheduler_set_sampler_state": (None, [VP, u64]),
    "tgh_scheduler_free": (None, [VP]),
    "tgh_sobol_matrices": (C.POINTER(u32), [C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
    "tgh_accel_build": (VP, [VP, VP, VP, C.c_uint32, C.c_char_p, C.c_size_t]),
    "tgh_accel_nodes": (VP, [VP, C.POINTER(C.c_uint32)]),
    "tgh_accel_wide_nodes": (VP, [VP, C.POINTER(C.c_uint32)]),
    "tgh_accel_free": (None, [VP]),
    "tgh_accel_build_instanced": (VP, [VP, VP, VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32, C.c_char_p, C.c_size_t]),
    "tgh_accel_recs": (VP, [VP, C.POINTER(C.c_uint32)]),
    "tgh_accel_tri_attrs": (VP, [VP]),
    "tgh_accel_inst_prims": (VP, [VP, C.POINTER(C.c_uint32)]),
    "tgh_accel_inst_leaf_boxes": (VP, [VP]),
    "tgh_accel_inst_tight_boxes": (VP, [VP]),
    "tgh_accel_counts": (None, [VP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "tgh_instance_tight_bounds": (None, [VP, C.c_uint32, C.c_uint32, VP, VP, VP, VP]),
    "tgh_scene_items": (C.c_uint32, [VP, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_int32))]),
    "tgh_top_tree_build": (C.c_int, [VP, C.c_uint32, VP, C.c_uint32]),
    "tgh_top_tree_for_scene": (C.c_int, [VP, C.c_uint32, VP, C.c_uint32, VP, C.c_uint32]),
    "tgh_leaf_bounds": (C.c_int, [VP, C.c_uint32, VP, VP]),
    "tgh_save_pfm": (C.c_int, [C.c_char_p, VP, C.c_int, C.c_int]),
    "tgh_load_hdr": (C.c_int, [C.c_char_p, VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}


def bind(lib):
    """Attach restype/argtypes for every declared symbol; raises AttributeError on a missing export."""
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load_library(path=None):
    path = path or os.environ.get("TUNGSTEN_AMD_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise ImportError(
            "tungsten_amd: native library %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make`). There is no Python/CPU fallback for the path tracer." % path)
    return bind(C.CDLL(path))
