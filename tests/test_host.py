"""Host side (C++11, no device needed): JSON scene loading, flattening and the BVH2 the device consumes
(the replacement of TraceableScene's Embree build, renderer/TraceableScene.hpp:57-137)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import scenes
import tungsten_amd as tg
from tungsten_amd import capi


def _np(ptr, n, dtype, cols):
    if n == 0:
        return np.zeros((0, cols), dtype)
    buf = (C.c_char*(n*cols*np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype).reshape(n, cols).copy()


def check_bvh(desc):
    """Every record in exactly one leaf, child boxes enclose their subtree, depth within the device limit."""
    nodes_f = _np(desc.nodes, desc.num_nodes, np.float32, 16)
    nodes_i = nodes_f.view(np.int32)
    recs = _np(desc.recs, desc.num_recs, np.float32, 12)
    seen = np.zeros(desc.num_recs, np.int32)

    def rec_bounds(first, count):
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for r in range(first, first + count):
            meta = recs[r].view(np.uint32)[3]
            kind = meta >> 29
            a, b, c = recs[r][0:3], recs[r][4:7], recs[r][8:11]
            if kind == 0:
                pts = [a, a + b, a + c]
            elif kind == 1:
                pts = [a, a + b, a + c, a + b + c]
            else:
                return None
            for p in pts:
                lo, hi = np.minimum(lo, p), np.maximum(hi, p)
        return lo, hi

    max_depth = 0
    stack = [(0, 1, None)]
    while stack:
        ref, depth, box = stack.pop()
        if ref < 0:
            first, count = ref & 0x07FFFFFF, (ref >> 27) & 15
            assert 1 <= count <= 15 and first + count <= desc.num_recs
            seen[first:first + count] += 1
            rb = rec_bounds(first, count)
            if rb is not None and box is not None:
                assert (rb[0] >= box[0] - 1e-4).all() and (rb[1] <= box[1] + 1e-4).all()
            continue
        max_depth = max(max_depth, depth)
        n = nodes_f[ref]
        lo0, hi0, lo1, hi1 = n[0:3], n[3:6], n[6:9], n[9:12]
        if box is not None:
            assert (np.minimum(lo0, lo1) >= box[0] - 1e-4).all() and (np.maximum(hi0, hi1) <= box[1] + 1e-4).all()
        stack.append((int(nodes_i[ref][12]), depth + 1, (lo0, hi0)))
        stack.append((int(nodes_i[ref][13]), depth + 1, (lo1, hi1)))
    assert (seen == 1).all()
    assert max_depth <= 48
    return max_depth


def test_cornell_flattening(tmp_path):
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(64, 36), spp=4))
    d = flat.desc.contents
    assert (flat.width, flat.height) == (64, 36)
    assert d.abi_version == 5
    assert d.num_objects == 8 and d.num_lights == 1 and d.num_infinite_lights == 0
    assert d.num_recs == 8                       # 6 quads + 2 cubes, analytic records (Quad.cpp / Cube.cpp)
    light = d.objects[d.lights[0]]
    assert light.type == 1 and light.emission >= 0 and light.light == 0
    assert np.allclose(list(d.textures[light.emission].value), [17, 12, 4])
    assert d.settings.max_bounces == 64 and d.settings.enable_light_sampling == 1 and d.settings.enable_two_sided_shading == 1
    assert d.camera.filter_type == 2 and d.camera.res_x == 64
    assert check_bvh(d) == flat.info.bvh_depth
    flat.close()


@pytest.mark.skipif(not scenes.have_materialtest(), reason="materialtest assets (oracle/_ref/data) not present")
def test_materialtest_flattening(tmp_path):
    flat = tg.FlattenedScene(scenes.materialtest(tmp_path, resolution=(64, 36), spp=4))
    d = flat.desc.contents
    assert d.num_recs == 80768 + 1               # three .wo3 meshes + the floor quad
    assert d.num_infinite_lights == 1 and d.num_lights == 1
    env = d.objects[d.infinite_lights[0]]
    tex = d.textures[env.emission]
    assert tex.type == 2 and (tex.w, tex.h) == (1024, 512) and tex.dist_offset >= 0
    # Distribution2D layout: marginalPdf[h] marginalCdf[h+1] pdf[w*h] cdf[(w+1)*h]
    assert d.num_dist_floats >= tex.dist_offset + 512 + 513 + 1024*512 + 1025*512
    dist = _np(d.dist, d.num_dist_floats, np.float32, 1)[:, 0][tex.dist_offset:]
    mcdf = dist[512:512 + 513]
    assert mcdf[0] == 0.0 and abs(mcdf[-1] - 1.0) < 1e-6 and (np.diff(mcdf) >= 0).all()
    depth = check_bvh(d)
    assert depth == flat.info.bvh_depth and depth <= 40
    flat.close()


def test_default_environment_is_injected_when_no_emitter(tmp_path):
    # TraceableScene.hpp:97-102: a scene without any light gets a white infinite sphere
    def strip(scene):
        for p in scene["primitives"]:
            p.pop("emission", None)
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, edit=strip))
    d = flat.desc.contents
    assert d.num_infinite_lights == 1
    flat.close()


def test_json_errors_are_reported(tmp_path):
    p = tmp_path/"bad.json"
    p.write_text("{ this is not json")
    with pytest.raises(tg.TungstenError):
        tg.FlattenedScene(str(p))
    def bad_bsdf(scene):
        scene["bsdfs"][0]["type"] = "hair"
    with pytest.raises(tg.TungstenError) as e:
        tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, edit=bad_bsdf, name="hair.json"))
    assert "hair" in str(e.value)


def test_bump_maps_are_refused_not_dropped(tmp_path):
    # Primitive::setupTangentFrame (Primitive.cpp:125-163): a varying bump texture perturbs the shading frame; a constant
    # one changes nothing (:130).  The first is outside this integrator's scope and must not render unperturbed.
    def checker_bump(scene):
        scene["bsdfs"][0]["bump"] = {"type": "checker", "on_color": 1.0, "off_color": 0.0, "res_u": 4, "res_v": 4}
    with pytest.raises(tg.TungstenError) as e:
        tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, edit=checker_bump, name="bump.json"))
    assert "bump" in str(e.value)
    def constant_bump(scene):
        scene["bsdfs"][0]["bump"] = 0.5
    tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, edit=constant_bump, name="bump_const.json")).close()


def test_pfm_round_trip(tmp_path):
    img = np.random.RandomState(1).rand(5, 7, 3).astype(np.float32)
    path = str(tmp_path/"a.pfm")
    assert tg.lib.tgh_save_pfm(path.encode(), img.ctypes.data, 7, 5) == 0
    assert (tg.load_pfm(path) == img).all()
