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
    assert d.abi_version == 10
    assert d.num_objects == 8 and d.num_lights == 1 and d.num_infinite_lights == 0
    assert d.num_recs == 8                       # 6 quads + 2 cubes, analytic records (Quad.cpp / Cube.cpp)
    light = d.objects[d.lights[0]]
    assert light.type == 1 and light.emission >= 0 and light.light == 0
    assert np.allclose(list(d.textures[light.emission].value), [17, 12, 4])
    assert d.settings.max_bounces == 64 and d.settings.enable_light_sampling == 1 and d.settings.enable_two_sided_shading == 1
    assert d.camera.filter_type == 2 and d.camera.res_x == 64
    assert check_bvh(d) == flat.info.bvh_depth
    flat.close()


@pytest.mark.skipif(not scenes.have_materialtest(), reason="materialtest assets (assets/) not present")
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


def test_bump_maps_reach_the_device_unless_constant(tmp_path):
    # Primitive::setupTangentFrame (Primitive.cpp:125-163): a varying bump texture sends the shading frame through the primitive's tangent
    # space (TgHipBsdf::bump1 = its texture index + 1); a constant one changes nothing (:128-131) and is dropped.
    def checker_bump(scene):
        scene["bsdfs"][0]["bump"] = {"type": "checker", "on_color": 1.0, "off_color": 0.0, "res_u": 4, "res_v": 4}
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, edit=checker_bump, name="bump.json"))
    d = flat.desc.contents
    bumped = [d.bsdfs[i].bump1 for i in range(d.num_bsdfs) if d.bsdfs[i].bump1]
    assert len(bumped) == 1 and d.textures[bumped[0] - 1].type == 1          # the checker
    flat.close()
    def constant_bump(scene):
        scene["bsdfs"][0]["bump"] = 0.5
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, edit=constant_bump, name="bump_const.json"))
    d = flat.desc.contents
    assert not any(d.bsdfs[i].bump1 for i in range(d.num_bsdfs))
    flat.close()


def test_pfm_round_trip(tmp_path):
    img = np.random.RandomState(1).rand(5, 7, 3).astype(np.float32)
    path = str(tmp_path/"a.pfm")
    assert tg.lib.tgh_save_pfm(path.encode(), img.ctypes.data, 7, 5) == 0
    assert (tg.load_pfm(path) == img).all()


def check_wide_bvh(desc):
    """Every record in exactly one leaf child; every child's quantised box encloses the exact box of what hangs below it;
    internal children consecutive, breadth first; depth within the device limit.  Returns (depth, nodes)."""
    n = desc.num_wide_nodes
    raw = np.frombuffer((C.c_char*(n*80)).from_address(C.addressof(desc.wide_nodes.contents)), np.uint8).reshape(n, 80).copy()
    origin = raw[:, 0:12].copy().view(np.float32).reshape(n, 3)
    exp, imask = raw[:, 12:15], raw[:, 15]
    child_base = raw[:, 16:20].copy().view(np.uint32)[:, 0]
    rec_base = raw[:, 20:24].copy().view(np.uint32)[:, 0]
    leaf_valid = raw[:, 24:28].copy().view(np.uint32)[:, 0]
    qlo, qhi = raw[:, 32:56].reshape(n, 3, 8), raw[:, 56:80].reshape(n, 3, 8)
    spacing = np.ldexp(np.float32(1.0), exp.astype(np.int32) - 127).astype(np.float32)
    recs = _np(desc.recs, desc.num_recs, np.float32, 12)
    seen = np.zeros(desc.num_recs, np.int32)

    def rec_box(r):
        kind = recs[r].view(np.uint32)[3] >> 29
        a, b, c = recs[r][0:3], recs[r][4:7], recs[r][8:11]
        pts = [a, a + b, a + c] + ([a + b + c] if kind == 1 else [])
        assert kind in (0, 1)
        return np.min(pts, axis=0), np.max(pts, axis=0)

    exact = {}           # node -> exact box of its subtree, filled bottom-up (children have larger indices)
    depth = np.zeros(n, np.int32)
    depth[0] = 1
    for i in range(n):
        assert depth[i] > 0
        k = 0
        for s in range(8):
            if imask[i] >> s & 1:
                c = child_base[i] + k
                assert c > i and c < n and depth[c] == 0
                depth[c] = depth[i] + 1
                k += 1
    for i in range(n - 1, -1, -1):
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        k = 0
        for s in range(8):
            dlo = origin[i] + qlo[i, :, s].astype(np.float32)*spacing[i]
            dhi = origin[i] + qhi[i, :, s].astype(np.float32)*spacing[i]
            if imask[i] >> s & 1:
                blo, bhi = exact[child_base[i] + k]
                k += 1
            else:
                bits = int(leaf_valid[i] >> (4*s)) & 15
                assert bits in (0, 1, 3, 7, 15)
                cnt, off = bin(bits).count("1"), bin(int(leaf_valid[i]) & ((1 << (4*s)) - 1)).count("1")
                if cnt == 0:
                    assert (qlo[i, :, s] == 255).all() and (qhi[i, :, s] == 0).all()     # empty slot: no ray passes
                    continue
                assert cnt <= 4
                first = rec_base[i] + off
                seen[first:first + cnt] += 1
                boxes = [rec_box(r) for r in range(first, first + cnt)]
                blo, bhi = np.min([b[0] for b in boxes], axis=0), np.max([b[1] for b in boxes], axis=0)
            tol = 2e-6*np.maximum(np.abs(blo), np.abs(bhi)) + 1e-7      # (records store v0, v1 - v0, v2 - v0: the corners re-round)
            assert (dlo <= blo + tol).all() and (dhi >= bhi - tol).all(), (i, s)
            # between 1/64 of a step and two steps of slack (WideBvh.cpp: no quantised plane coincides with the box it bounds)
            assert (blo - dlo <= spacing[i]*2.001 + 1e-6*np.abs(blo)).all() and (dhi - bhi <= spacing[i]*2.001 + 1e-6*np.abs(bhi)).all()
            assert (blo - dlo >= spacing[i]/64.5 - tol).all() and (dhi - bhi >= spacing[i]/64.5 - tol).all()
            lo, hi = np.minimum(lo, blo), np.maximum(hi, bhi)
        exact[i] = (lo, hi)
    assert (seen == 1).all()
    assert depth.max() <= 32
    return int(depth.max()), n


@pytest.mark.skipif(not scenes.have_materialtest(), reason="materialtest assets (assets/) not present")
def test_wide_bvh_is_a_conservative_collapse_of_the_bvh2(tmp_path):
    import oracle_lib
    flat = tg.FlattenedScene(scenes.materialtest(tmp_path, resolution=(64, 36), spp=4))
    d = flat.desc.contents
    assert d.num_wide_nodes > 0
    depth, n = check_wide_bvh(d)
    assert n < d.num_nodes//3 and depth <= 12          # 80 768 BVH2 nodes collapse to well under a third
    check_bvh(d)                                        # the BVH2 still describes the re-ordered records
    # both walks find the same closest hits (ties between coincident / edge-sharing records aside): camera rays, then rays
    # leaving the surfaces they hit
    rays = []
    for y in range(36):
        for x in range(64):
            o, dd = oracle_lib.camera_ray(flat.desc, x, y, 0.5, 0.5)
            rays.append(list(o) + [1e-4] + list(dd) + [np.inf])
    rays = np.array(rays, np.float32)
    rays[0:3, 4:7] = [[1, 0, 0], [0, -1, 0], [0, 0, 1]]       # axis-parallel directions
    h = oracle_lib.trace_rays(flat.desc, rays)[0]
    hit = h["rec"] >= 0
    p = rays[hit, 0:3] + rays[hit, 4:7]*h["t"][hit, None]
    dirs = np.random.RandomState(3).randn(len(p), 3)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.concatenate([rays, np.concatenate([p, np.full((len(p), 1), 5e-4), dirs, np.full((len(p), 1), np.inf)], axis=1).astype(np.float32)])
    h2, n2, p2 = oracle_lib.trace_rays(flat.desc, rays)
    hw, nw, pw = oracle_lib.trace_rays(flat.desc, rays, wide=True)
    same = h2["rec"] == hw["rec"]
    assert same.mean() >= 0.999
    assert (h2["t"][same] == hw["t"][same]).all() and (h2["u"][same] == hw["u"][same]).all()
    assert nw < 0.45*n2 and pw < 1.6*p2                 # ~6 node visits per ray instead of ~15, a few more record tests
    flat.close()


def test_flat_list_scenes_carry_no_wide_bvh_and_instanced_scenes_the_references_tree(tmp_path):
    """Scenes with an `instances` primitive (ABI 8): the scene's BVH2 holds ONE record for the primitive, behind which the reference's own
    tree over the instances follows (csrc/host/RefInstanceBvh.cpp restates its builder; stored as BVH2 nodes with the reference's exact
    child boxes, leaves of one or two instances through inst_prims) -- what closest hits walk, in the reference's order --, and the wide
    BVH holds the instance records themselves, each boxed by its leaf of that tree -- what any-hit queries walk."""
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1))
    assert flat.desc.contents.num_wide_nodes == 0
    flat.close()
    flat = tg.FlattenedScene(scenes.instances10k(tmp_path, resolution=(16, 9), spp=1, count=300, n_lat=12, n_lon=12))
    d = flat.desc.contents
    assert d.num_instances == 300 and d.num_wide_nodes > 4 and d.num_inst_prims == 300
    recs = _np(d.recs, d.num_recs, np.float32, 12).view(np.uint32)
    kinds = recs[:d.num_top_recs, 3] >> 29
    assert (kinds == 4).sum() == 300 and (kinds == 7).sum() == 1 and kinds[d.num_top_recs - 1] == 7      # the set record follows the wide BVH's records
    prims = _np(d.inst_prims, d.num_inst_prims, np.uint32, 1).reshape(-1)
    assert sorted(prims.tolist()) == sorted(np.nonzero(kinds == 4)[0].tolist())                          # every instance in exactly one leaf slot
    wide_roots = set()
    for r in np.nonzero(kinds == 4)[0]:
        assert 0 < recs[r][10] < d.num_wide_nodes and recs[r][9] < d.num_inst_prims                      # c[2]: the master's wide root, c[1]: its leaf's first slot
        wide_roots.add(int(recs[r][10]))
    assert len(wide_roots) == 4
    # the reference's tree behind the set record: every inner node's two boxes contain the boxes of everything below them (the builder
    # computes them as unions), a leaf holds one or two instances, and every instance's position lies inside its leaf's box
    nodes = _np(d.nodes, d.num_nodes, np.float32, 16)
    refs = nodes.view(np.int32)
    fl = recs.view(np.float32)
    boxes = _np(d.inst_leaf_boxes, d.num_inst_prims, np.float32, 8)
    root = int(recs[d.num_top_recs - 1].view(np.int32)[8])
    assert 0 < root < d.num_nodes
    seen = []

    def walk(ref, lo, hi, depth):
        assert depth < 64
        if ref < 0:
            first, count = ref & 0x07FFFFFF, (ref >> 27) & 15
            assert 1 <= count <= 2
            for k in range(first, first + count):
                ri = int(prims[k])
                seen.append(ri)
                assert (fl[ri][0:3] >= lo - 1e-4).all() and (fl[ri][0:3] <= hi + 1e-4).all()
                assert recs[ri][9] == first
                leaf = boxes[first]
                assert (leaf[0:3] == lo).all() and (leaf[4:7] == hi).all()
            return lo, hi
        n = nodes[ref]
        l0, h0, l1, h1 = n[0:3], n[3:6], n[6:9], n[9:12]
        for (cl, ch), child in (((l0, h0), int(refs[ref][12])), ((l1, h1), int(refs[ref][13]))):
            assert (cl >= lo).all() and (ch <= hi).all()
            walk(child, cl, ch, depth + 1)
        return lo, hi
    walk(root, fl[d.num_top_recs - 1][0:3], fl[d.num_top_recs - 1][4:7], 0)
    assert sorted(seen) == sorted(prims.tolist())
    flat.close()


def _bitmap_of(desc, w, h):
    """texels (float32 [h, w, c]) of the first bitmap texture of that size in the flattened scene"""
    d = desc.contents
    for i in range(d.num_textures):
        t = d.textures[i]
        if t.type == 2 and t.w == w and t.h == h:
            c = 3 if (t.flags & 4) else 1
            buf = (C.c_float*(w*h*c)).from_address(C.addressof(d.texels.contents) + 4*t.texel_offset)
            return np.frombuffer(buf, np.float32).reshape(h, w, c).copy()
    raise AssertionError("no %dx%d bitmap in the scene" % (w, h))


def test_png_textures_decode_to_the_floats_of_the_reference_lookup(tmp_path):
    """8-bit textures (.png; the reference: lodepng -> ImageIO::loadLdr -> BitmapTexture::getRgb, io/ImageIO.cpp:493-526,
    textures/BitmapTexture.cpp:139-154): the host's own inflate + PNG decoder against images written here by zlib, for every colour
    type / bit depth / scanline filter / block type it reads; RGB requests go through the 2.2 table floor(255 (i/255)^2.2)."""
    rs = np.random.RandomState(3)
    w, h = 37, 23
    lut = np.floor(255.0*(np.arange(256)/255.0)**2.2).astype(np.uint8)
    to_f = lambda a: a.astype(np.float32)*np.float32(1.0/255.0)
    smooth = (np.add.outer(np.arange(h)*5, np.arange(w)*3)[..., None] + np.array([0, 40, 90])) % 256      # compressible: dynamic codes, long matches
    noise = rs.randint(0, 256, (h, w, 3))
    cases = []
    for name, img, level, filters, chunks in (("smooth9", smooth, 9, (0, 1, 2, 3, 4), 1), ("noise6", noise, 6, (4, 3), 3), ("stored", noise, 0, (0,), 1),
                                                ("fixed", smooth, 1, (1,), 2)):
        path = str(tmp_path/(name + ".png"))
        scenes.write_png(path, img, 2, filters=filters, level=level, idat_chunks=chunks)
        cases.append((path, to_f(lut[img]), {}))
    # RGBA: alpha is dropped; gamma_correct off keeps the raw bytes
    rgba = rs.randint(0, 256, (h, w, 4))
    scenes.write_png(str(tmp_path/"rgba.png"), rgba, 6)
    cases.append((str(tmp_path/"rgba.png"), to_f(rgba[..., :3]), {"gamma_correct": False}))
    # grey (8 / 4 / 1 bit), grey + alpha, palette (8 / 2 bit), 16-bit RGB (the high byte counts)
    g8 = rs.randint(0, 256, (h, w))
    scenes.write_png(str(tmp_path/"g8.png"), g8, 0)
    cases.append((str(tmp_path/"g8.png"), to_f(lut[np.repeat(g8[..., None], 3, -1)]), {}))
    g4 = rs.randint(0, 16, (h, w))
    scenes.write_png(str(tmp_path/"g4.png"), g4, 0, depth=4)
    cases.append((str(tmp_path/"g4.png"), to_f(lut[np.repeat((g4*255//15)[..., None], 3, -1)]), {}))
    g1 = rs.randint(0, 2, (h, w))
    scenes.write_png(str(tmp_path/"g1.png"), g1, 0, depth=1)
    cases.append((str(tmp_path/"g1.png"), to_f(lut[np.repeat((g1*255)[..., None], 3, -1)]), {}))
    ga = rs.randint(0, 256, (h, w, 2))
    scenes.write_png(str(tmp_path/"ga.png"), ga, 4)
    cases.append((str(tmp_path/"ga.png"), to_f(lut[np.repeat(ga[..., :1], 3, -1)]), {}))
    pal = rs.randint(0, 256, (256, 3))
    idx = rs.randint(0, 256, (h, w))
    scenes.write_png(str(tmp_path/"p8.png"), idx, 3, palette=pal)
    cases.append((str(tmp_path/"p8.png"), to_f(lut[pal[idx]]), {}))
    idx2 = rs.randint(0, 4, (h, w))
    scenes.write_png(str(tmp_path/"p2.png"), idx2, 3, depth=2, palette=pal[:4])
    cases.append((str(tmp_path/"p2.png"), to_f(lut[pal[idx2]]), {}))
    rgb16 = rs.randint(0, 65536, (h, w, 3))
    scenes.write_png(str(tmp_path/"rgb16.png"), rgb16, 2, depth=16)
    cases.append((str(tmp_path/"rgb16.png"), to_f(lut[rgb16 >> 8]), {}))
    for k, (png, expect, extra) in enumerate(cases):
        def edit(scene, png=png, extra=extra):
            scene["bsdfs"].append({"name": "tex", "type": "lambert", "albedo": dict({"type": "bitmap", "file": os.path.basename(png)}, **extra)})
            scene["primitives"][0]["bsdf"] = "tex"
        flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, name="png%d.json" % k, edit=edit))
        got = _bitmap_of(flat.desc, w, h)
        flat.close()
        assert got.shape == expect.shape and (got == expect).all(), os.path.basename(png)
    # scalar requests: the integer channel average without gamma (roughness, REQUEST_AVERAGE); the alpha channel for a transparency
    # bsdf's "alpha" (REQUEST_AUTO: a .png always reports four channels, io/ImageIO.cpp:386-407)
    def edit_scalar(scene):
        scene["bsdfs"].append(dict({"name": "rc", "type": "rough_conductor", "albedo": 1, "roughness": "rgba.png"}, **scenes._CU))
        scene["bsdfs"].append({"name": "cut", "type": "transparency", "base": {"type": "lambert", "albedo": 0.5}, "alpha": "rgba.png"})
        scene["primitives"][0]["bsdf"] = "rc"
        scene["primitives"][1]["bsdf"] = "cut"
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, name="scalar.json", edit=edit_scalar))
    d = flat.desc.contents
    scalars = []
    for i in range(d.num_textures):
        t = d.textures[i]
        if t.type == 2 and (t.w, t.h) == (w, h) and not (t.flags & 4):
            buf = (C.c_float*(w*h)).from_address(C.addressof(d.texels.contents) + 4*t.texel_offset)
            scalars.append(np.frombuffer(buf, np.float32).reshape(h, w).copy())
    flat.close()
    avg = to_f((rgba[..., 0] + rgba[..., 1] + rgba[..., 2])//3)
    assert len(scalars) == 2
    assert any((a == avg).all() for a in scalars) and any((a == to_f(rgba[..., 3])).all() for a in scalars)
    # a truncated file is an error, not a black texture
    bad = str(tmp_path/"bad.png")
    open(bad, "wb").write(open(cases[0][0], "rb").read()[:200])
    with pytest.raises(Exception):
        tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, name="bad.json",
                                         edit=lambda s: (s["bsdfs"].append({"name": "tex", "type": "lambert", "albedo": os.path.basename(bad)}), s["primitives"][0].update(bsdf="tex"))))


def test_pfm_textures(tmp_path):
    """.pfm bitmaps (io/ImageIO.cpp:298-338): rows bottom to top; a scalar file feeds all three channels of an RGB request."""
    rs = np.random.RandomState(9)
    w, h = 19, 11
    rgb = rs.rand(h, w, 3).astype(np.float32)*4
    grey = rs.rand(h, w).astype(np.float32)
    def write(path, a):
        with open(path, "wb") as f:
            f.write(b"PF\n" if a.ndim == 3 else b"Pf\n")
            f.write(("%d %d\n-1.0\n" % (w, h)).encode())
            f.write(np.ascontiguousarray(a[::-1]).tobytes())
    write(str(tmp_path/"c.pfm"), rgb)
    write(str(tmp_path/"g.pfm"), grey)
    for k, (name, expect) in enumerate((("c.pfm", rgb), ("g.pfm", np.repeat(grey[..., None], 3, -1)))):
        def edit(scene, name=name):
            scene["bsdfs"].append({"name": "tex", "type": "lambert", "albedo": name})
            scene["primitives"][0]["bsdf"] = "tex"
        flat = tg.FlattenedScene(scenes.cornell(tmp_path, resolution=(16, 9), spp=1, name="pfm%d.json" % k, edit=edit))
        got = _bitmap_of(flat.desc, w, h)
        flat.close()
        assert (got == expect).all(), name


@pytest.mark.parametrize("case", ["cornell_skydome", "cornell_skydome_alien"])
def test_skydome_image_is_the_image_the_reference_bakes(case, tmp_path):
    """The `skydome` primitive: the host's restatement of the Hosek-Wilkie sky model and of Skydome::prepareForRender
    (tungsten_amd/csrc/host/SkyModel.cpp over tungsten_amd/data/skydome_tables.bin) bakes the 512 x 256 image the reference bakes for
    the same transform / temperature / turbidity / intensity -- bit for bit on every 4th row and 8th column (tests/golden/*_sky.npz,
    dumped from the reference's own Skydome by `ref_harness sky-image`), with the same sum over all texels -- and flattens it to an
    unrotated, sampled (or not) infinite sphere with the spherical Distribution2D."""
    mk, kw = scenes.GOLDEN_CASES[case]
    path = mk(tmp_path, **kw)
    gold = np.load(os.path.join(scenes.GOLDEN, case + "_sky.npz"))
    flat = tg.FlattenedScene(path)
    d = flat.desc.contents
    sky = [d.objects[i] for i in range(d.num_objects) if d.objects[i].flags & 4]
    assert len(sky) == 1 and sky[0].type == 4                                         # TGHIP_OBJF_SKYDOME on a TGHIP_OBJ_INFINITE_SPHERE
    t = d.textures[sky[0].emission]
    assert (t.w, t.h) == (512, 256) and t.flags & 1 and not t.flags & 2                # interpolated, not clamped
    tex = np.ctypeslib.as_array(d.texels, (d.num_texel_floats,))[t.texel_offset:t.texel_offset + 512*256*3].reshape(256, 512, 3)
    assert (tex[::4, ::8] == gold["sub"]).all()
    assert float(tex.astype(np.float64).sum()) == float(gold["total"])
    assert (tex[130:] == 0).all() and (tex[128] == tex[127]).all() and (tex[129] == tex[127]).all()   # Skydome.cpp:302-303
    sampled = case == "cornell_skydome"
    assert bool(sky[0].flags & 2) == sampled and (t.dist_offset >= 0) == sampled and (sky[0].light >= 0) == sampled
    flat.close()


def test_bitmap_aperture_distribution(tmp_path):
    """A thin-lens camera's bitmap aperture arrives as the Distribution2D of BitmapTexture::makeSamplable(MAP_UNIFORM) over the image's
    grey levels (no sin(theta) row weights; the 3 x 3 dilation of BitmapTexture.cpp:413-428): marginals and rows are normalised, texels
    next to the ring carry its weight, the far corners none."""
    mk, kw = scenes.GOLDEN_CASES["cornell_thinlens_bitmap"]
    flat = tg.FlattenedScene(mk(tmp_path, **kw))
    d = flat.desc.contents
    cam = d.camera
    assert cam.type == 1 and cam.aperture_type == 2 and (cam.aperture_w, cam.aperture_h) == (24, 20)
    dist = np.ctypeslib.as_array(d.dist, (d.num_dist_floats,))[cam.aperture_dist:]
    w, h = 24, 20
    mpdf, mcdf, pdf, cdf = dist[:h], dist[h:2*h + 1], dist[2*h + 1:2*h + 1 + w*h].reshape(h, w), dist[2*h + 1 + w*h:2*h + 1 + w*h + (w + 1)*h].reshape(h, w + 1)
    assert abs(mcdf[-1] - 1.0) < 1e-6 and mcdf[0] == 0 and (np.diff(mcdf) >= 0).all()
    assert np.allclose(cdf[:, -1], 1.0, atol=1e-6) and (cdf[:, 0] == 0).all()
    assert np.allclose(mpdf.sum(), 1.0, atol=1e-5) and np.allclose(pdf.sum(axis=1), 1.0, atol=1e-5)
    assert pdf[10, 12] == 0 and pdf[3, 11] > pdf[10, 3] > 0        # the hole of the ring; the notch is the brightest spot
    flat.close()


def test_wide_walk_loses_no_hit_on_grid_aligned_geometry(tmp_path):
    """Grazing and corner rays on axis-aligned tiles at exact grid coordinates, 4096 units from the origin (tests/scenes.py: tile_terraces,
    terrace_rays): the wide walk -- quantised planes, distances by fma(q, spacing/d, (origin - o)/d) -- returns the hits of the BVH2 walk, whose
    boxes are the exact ones: same record or, at a shared edge or corner, a record at the same distance."""
    import oracle_lib
    flat = tg.FlattenedScene(scenes.tile_terraces(tmp_path))
    d = flat.desc.contents
    assert d.num_wide_nodes > 0
    check_wide_bvh(d)
    rays = scenes.terrace_rays()
    wide = oracle_lib.trace_rays(flat.desc, rays, wide=True)[0]
    bvh2 = oracle_lib.trace_rays(flat.desc, rays)[0]
    flat.close()
    lost = (wide["rec"] < 0) & (bvh2["rec"] >= 0)
    assert not lost.any(), "the wide walk lost %d hits" % int(lost.sum())
    # (the other way round happens: a ray through an exact grid point has 0 * inf = NaN in the BVH2 walk's exact-box slab test and is
    # culled there; the wide walk keeps 1/d finite and its planes have slack -- a quarter of the rays of this adversarial set)
    assert ((wide["rec"] >= 0) & (bvh2["rec"] < 0)).mean() < 0.4
    both = (wide["rec"] >= 0) & (bvh2["rec"] >= 0)
    assert both.mean() > 0.5
    # the same distance; where two triangles share the edge or corner that was hit either may win (their distances differ by an ulp)
    assert np.allclose(wide["t"][both], bvh2["t"][both], rtol=3e-7, atol=0)
    assert (wide["rec"][both] == bvh2["rec"][both]).mean() > 0.99


def test_image_readers_refuse_oversized_and_overlong_input(tmp_path):
    """Bitmaps come from scene files: a header that announces 2^31 pixels or a deflate stream that unpacks to more than the header's
    image (a zip bomb) is refused with a message, not decoded."""
    import struct
    import zlib

    def png(w, h, payload):
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + chunk(b"IDAT", payload) + chunk(b"IEND", b"")

    def load_error(name, data):
        p = tmp_path/name
        p.write_bytes(data)
        scene = scenes.cornell(tmp_path, name=name + ".json", edit=lambda s: s["bsdfs"][0].update(albedo=name))
        with pytest.raises(tg.TungstenError) as e:
            tg.FlattenedScene(scene)
        return str(e.value)

    assert "out of range" in load_error("huge.png", png(70000, 70000, zlib.compress(b"\0"*64)))
    assert "more image data" in load_error("bomb.png", png(4, 4, zlib.compress(b"\0"*(1 << 22))))
    assert "bad PFM header" in load_error("huge.pfm", b"PF\n70000 70000\n-1.0\n" + b"\0"*64)
    ok = tmp_path/"ok.png"
    ok.write_bytes(png(4, 4, zlib.compress(b"".join(b"\0" + bytes([40*y + 10*x for x in range(4)]) for y in range(4)))))
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, name="ok.json", edit=lambda s: s["bsdfs"][0].update(albedo="ok.png")))
    flat.close()


def _libm_host():
    import ctypes as C
    if "fma" not in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        pytest.skip("host CPU without FMA3: glibc runs its non-FMA sinf / cosf / expf variants here")
    path = os.path.join(scenes.ROOT, "oracle", "libm_host.so")
    if not os.path.exists(path):
        pytest.skip("oracle/libm_host.so not built")
    lib = C.CDLL(path)
    lib.libm_host_sweep.restype = C.c_ulonglong
    lib.libm_host_sweep.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint]
    lib.libm_host_sweep2.restype = C.c_ulonglong
    lib.libm_host_sweep2.argtypes = [C.c_int, C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
    for f in (lib.libm_host_eval, lib.libm_host_ref):
        f.restype = None
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    return lib


def test_libm_restatements_match_the_host_libm():
    """csrc/hip/pt_libm.h -- glibc's sinf / cosf / logf / expf (and atanf / atan2f / powf / cbrtf / tanf) restated for the kernels -- compiled for the host (oracle/libm_host.cpp) against
    the image's libm, bit for bit: every 5th float of either sign (the full sweep, stride 1, is `python tools/libm_sweep.py`: zero
    mismatches over all 2^32 bit patterns inside the functions' ranges), and the array entry points the GPU test uses."""
    lib = _libm_host()
    for fn in (0, 1, 2, 3, 4, 5, 7, 8, 12):       # sinf, cosf, logf, expf, sincos (sin), sincos (cos), atanf, cbrtf, tanf (|x| < 120)
        for lo, hi in ((0x00000000, 0x7F800000), (0x80000000, 0xFF800000)):
            assert lib.libm_host_sweep(fn, lo + fn, hi, 5) == 0, (fn, hex(lo))
    # the two-argument ones: atan2f and powf on 2 x 10^7 pseudo-random pairs each
    for fn in (0, 1):
        tested = C.c_ulonglong(0)
        assert lib.libm_host_sweep2(fn, 20000000, 7, C.byref(tested)) == 0 and tested.value > 8000000, fn
    rng = np.random.default_rng(5)
    for fn, x in ((0, rng.random(200000)*6.2831855), (1, rng.random(200000)*6.2831855), (4, rng.random(200000)*3.1415927), (5, -rng.random(200000)*100),
                  (2, 1.0 - rng.random(200000)), (2, rng.random(200000)*1e30), (3, -rng.random(200000)*80), (3, rng.random(200000)*80)):
        x = np.ascontiguousarray(x, np.float32)
        got, want = np.empty_like(x), np.empty_like(x)
        lib.libm_host_eval(fn, x.ctypes.data, got.ctypes.data, x.size)
        lib.libm_host_ref(fn, x.ctypes.data, want.ctypes.data, x.size)
        ok = ~np.isnan(got)                     # (sin / cos outside |x| < 120: the host build answers NaN, the device folds the angle)
        assert ok.mean() > 0.99 and (got[ok].view(np.uint32) == want[ok].view(np.uint32)).all(), fn


def test_double_libm_restatements_match_the_host_libm():
    """pt_libm.h: expD / logD / erfD -- glibc 2.35's double-precision exp, log and erf as AtmosphericMedium::inverseOpticalDepth calls them (media/AtmosphericMedium.cpp:
    113-122, math/Erf.hpp:192-245), the fused operations of its FMA builds spelt out -- compiled for the host against the image's libm, bit for bit, on 3 x 10^7
    pseudo-random arguments each: arbitrary bit patterns (special cases, subnormal results, overflow), the functions' ranges, tiny arguments, the call sites' ranges."""
    import ctypes as C
    lib = _libm_host()
    lib.libm_host_sweepd.restype = C.c_ulonglong
    lib.libm_host_sweepd.argtypes = [C.c_int, C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
    for fn in (0, 1, 2):
        tested = C.c_ulonglong(0)
        assert lib.libm_host_sweepd(fn, 30000000, 5 + fn, C.byref(tested)) == 0 and tested.value >= 29999000, fn
    # the array entry points the GPU test compares the device with
    x = np.array([-745.2, -708.5, -1e-20, 0.3, 1.0, 709.7, 710.0, -800.0, np.inf, -np.inf, np.nan, 5e-324, 0.96, 1.04, 0.25, 3.7], np.float64)
    for fn in (0, 1, 2):
        got, want = np.empty_like(x), np.empty_like(x)
        lib.libm_host_evald(fn, x.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
        lib.libm_host_refd(fn, x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
        assert ((got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))).all(), fn


def test_oracle_rcpps_is_the_intel_instruction():
    """oracle/oracle.c: intel_rcpps -- the rule behind Embree's rcp() in its triangle test (simd/vfloat4_sse2.h:166-173: RCPPS + one Newton step),
    restated in integer arithmetic because the instruction's result is the vendor's, not IEEE's -- against the instruction itself on a host that
    has Intel's: every exponent x every one of the 2048 table indices x mantissa bits below the index, specials, and 2 M random bit patterns
    (tools/rcpps_sweep.c runs all 2^32: profiles/r4_rcpps_sweep.txt).  Elsewhere (AMD hosts: another RCPPS) only the rule's own properties are checked."""
    import ctypes as C
    import oracle_lib
    rng = np.random.default_rng(3)
    e = np.arange(0, 256, dtype=np.uint32)[:, None, None] << 23
    i = np.arange(0, 2048, dtype=np.uint32)[None, :, None] << 12
    low = np.array([0, 1, 0x7ff, 0xfff], np.uint32)[None, None, :]
    bits = np.concatenate([(e | i | low).reshape(-1), (e | i | low).reshape(-1) | np.uint32(0x80000000),
                           rng.integers(0, 1 << 32, 1 << 21, dtype=np.uint64).astype(np.uint32)])
    x = np.ascontiguousarray(bits.view(np.float32))
    est, full = np.empty_like(x), np.empty_like(x)
    oracle_lib._lib.oracle_embree_rcp(1, x.ctypes.data_as(C.c_void_p), est.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    oracle_lib._lib.oracle_embree_rcp(0, x.ctypes.data_as(C.c_void_p), full.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    # the rule's own properties: |relative error| of the estimate <= 1.5 * 2^-12 (Intel's bound for RCPPS), of estimate + Newton step <= 2^-22
    normal = np.isfinite(x) & (np.abs(x) > 1e-30) & (np.abs(x) < 1e30)
    xd = x[normal].astype(np.float64)
    assert np.abs(est[normal]*xd - 1.0).max() <= 1.5*2.0**-12
    assert np.abs(full[normal]*xd - 1.0).max() <= 2.0**-22
    if "GenuineIntel" not in open("/proc/cpuinfo").read():
        pytest.skip("the instruction is checked on Intel hosts only (AMD's RCPPS is a different function)")
    path = os.path.join(scenes.ROOT, "oracle", "libm_host.so")
    if not os.path.exists(path):
        pytest.skip("oracle/libm_host.so not built")
    lib = C.CDLL(path)
    hw = np.empty_like(x)
    lib.libm_host_rcpps_hw(x.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    same = (hw.view(np.uint32) == est.view(np.uint32))
    assert same.all(), (int((~same).sum()), hex(int(x.view(np.uint32)[~same][0])), hex(int(hw.view(np.uint32)[~same][0])), hex(int(est.view(np.uint32)[~same][0])))


def test_leaf_bounds_are_the_oracles(tmp_path):
    """Quad::bounds / Cube::bounds / Sphere::bounds as the library restates them for the items of the reference's top-level tree
    (include/tungsten_host.h: tgh_leaf_bounds) against oracle.c's restatement, bit for bit, on scenes with rotated cubes and spheres; a
    record kind without restated bounds (triangles) answers 0 on both sides.  (tests/test_top_tree.py holds the library's to the reference's own.)"""
    import oracle_lib
    lib = capi.load_library()
    answers = {}
    cases = [(scenes.cornell, {}), (scenes.cornell_zoo, {"which": "zoo_d"}), scenes.GOLDEN_CASES["cornell_disks"], scenes.GOLDEN_CASES["cornell_cylinders"],
             scenes.GOLDEN_CASES["cornell_bump"]]
    for mk, kw in cases:
        flat = tg.FlattenedScene(mk(tmp_path, **dict(kw, resolution=(16, 9), spp=1)))
        d = flat.desc.contents
        a = np.zeros((2, 3), np.float32)
        for i in range(min(d.num_recs, 64)):
            kind, obj = d.recs[i].meta >> 29, d.recs[i].meta & 0x1FFFFFFF
            ra = lib.tgh_leaf_bounds(C.byref(d.objects[obj]), kind, a[0].ctypes.data, a[1].ctypes.data)
            b = oracle_lib.leaf_bounds(flat.desc, i)
            assert ra == (1 if kind in (1, 2, 3, 5, 6) else 0) and (b is not None) == bool(ra), (kind, ra)
            answers[kind] = ra
            if ra:
                assert (a[0].view(np.uint32) == b[0].view(np.uint32)).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all() and (a[0] <= a[1]).all()
        flat.close()
    assert answers == {0: 0, 1: 1, 2: 1, 3: 1, 5: 1, 6: 1}    # triangles, quads, cubes, spheres, disks, cylinders
