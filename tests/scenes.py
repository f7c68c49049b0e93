"""Scene fixtures for the tests: variants of the committed Cornell box and, when its assets are
available (assets/ populated by __graft_entry__.build() from the reference tree), of materialtest.
The benchmark's own workloads live in tungsten_amd/workloads.py and are re-exported here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tungsten_amd.workloads import (CORNELL, MATERIALTEST_DIR, ASSETS, _CU, _CHECKER, _mt_material, cornell, displaced_sphere,  # noqa: E402,F401
                                    have_materialtest, instances10k, materialtest, mesh1m, variant, write_wo3)
VOLUMETRIC_CAUSTIC = os.path.join(ROOT, "scenes", "volumetric-caustic", "scene.json")
NON_EXPONENTIAL = os.path.join(ROOT, "scenes", "non-exponential", "scene.json")
GOLDEN = os.path.join(ROOT, "tests", "golden")


WATER_DIR = os.path.join(ASSETS, "water-caustic")


def have_water_caustic():
    return os.path.exists(os.path.join(WATER_DIR, "scene.json"))


def water_caustic(tmpdir, **kw):
    """data/example-scenes/water-caustic of the reference (Cornell box + smooth 67 000-triangle dielectric water surface),
    rendered with the path tracer instead of its shipped progressive photon mapper (same max_bounces = 8)."""
    for f in os.listdir(WATER_DIR):
        if f.endswith(".json"):
            continue
        link = os.path.join(str(tmpdir), f)
        if not os.path.exists(link):
            os.symlink(os.path.join(WATER_DIR, f), link)
    user = kw.pop("edit", None)

    def edit(scene):
        keep = {k: scene["integrator"][k] for k in ("min_bounces", "max_bounces", "enable_consistency_checks", "enable_two_sided_shading")}
        scene["integrator"] = dict(keep, type="path_tracer", enable_light_sampling=True)
        if user:
            user(scene)
    return variant(os.path.join(WATER_DIR, "scene.json"), str(tmpdir), kw.pop("name", "water_caustic.json"), edit=edit, **kw)


# ---- BSDF "zoo": Cornell box variants that exercise every BSDF/texture type on the hot path ----

ZOO = {
    "zoo_a": {
        "shortBox": {"type": "dielectric", "ior": 1.5, "albedo": 1},
        "tallBox": {"type": "mirror", "albedo": [0.9, 0.9, 0.95]},
        "floor": {"type": "lambert", "albedo": _CHECKER},
        "leftWall": dict({"type": "rough_conductor", "distribution": "ggx", "roughness": 0.2, "albedo": 1}, **_CU),
        "backWall": {"type": "plastic", "ior": 1.5, "thickness": 1.0, "sigma_a": [0.2, 0.4, 0.1], "albedo": [0.3, 0.5, 0.7]},
    },
    "zoo_b": {
        "shortBox": {"type": "rough_dielectric", "ior": 1.45, "distribution": "beckmann", "roughness": 0.15, "albedo": 1},
        "tallBox": dict({"type": "conductor", "albedo": [0.95, 0.9, 0.8]}, **_CU),
        "floor": {"type": "rough_plastic", "ior": 1.6, "thickness": 0.5, "sigma_a": 0.3, "distribution": "ggx",
                  "roughness": 0.25, "albedo": [0.6, 0.3, 0.2]},
        "rightWall": {"type": "smooth_coat", "ior": 1.4, "thickness": 2.0, "sigma_a": [0.3, 0.1, 0.05], "albedo": 1,
                      "substrate": dict({"type": "rough_conductor", "distribution": "beckmann", "roughness": 0.3, "albedo": 1}, **_CU)},
        "backWall": {"type": "mixed", "ratio": 0.4, "albedo": 1,
                     "bsdf0": {"type": "lambert", "albedo": [0.7, 0.7, 0.2]}, "bsdf1": {"type": "mirror", "albedo": 0.9}},
        "leftWall": {"type": "transparency", "alpha": 0.6, "albedo": 1, "base": {"type": "lambert", "albedo": [0.63, 0.065, 0.05]}},
    },
    "zoo_c": {
        "shortBox": {"type": "rough_dielectric", "ior": 1.7, "distribution": "phong", "roughness": 0.3, "enable_refraction": False, "albedo": 1},
        "tallBox": dict({"type": "rough_conductor", "distribution": "phong", "roughness": 0.15, "albedo": [0.9, 0.8, 0.7]}, **_CU),
        "floor": {"type": "transparency", "alpha": _CHECKER, "albedo": 1, "base": {"type": "lambert", "albedo": 0.7}},
        "ceiling": {"type": "mixed", "ratio": _CHECKER, "albedo": 1,
                    "bsdf0": dict({"type": "rough_conductor", "distribution": "ggx", "roughness": 0.4, "albedo": 1}, **_CU),
                    "bsdf1": {"type": "lambert", "albedo": [0.2, 0.5, 0.8]}},
        "backWall": {"type": "dielectric", "ior": 1.33, "enable_refraction": False, "albedo": 1},
        "rightWall": {"type": "rough_plastic", "ior": 1.3, "thickness": 1.0, "sigma_a": 0.0, "distribution": "beckmann",
                      "roughness": 0.1, "albedo": [0.14, 0.45, 0.091]},
    },
    # round 4: the five remaining non-fibre types of bsdfs/BsdfFactory.cpp:29-51
    "zoo_e": {
        "leftWall": {"type": "oren_nayar", "roughness": 0.6, "albedo": [0.63, 0.065, 0.05]},
        "rightWall": {"type": "phong", "exponent": 40.0, "diffuse_ratio": 0.3, "albedo": [0.14, 0.45, 0.091]},
        "backWall": {"type": "rough_coat", "ior": 1.4, "thickness": 1.0, "sigma_a": [0.2, 0.1, 0.3], "distribution": "ggx", "roughness": 0.2, "albedo": 1,
                     "substrate": {"type": "lambert", "albedo": [0.5, 0.6, 0.7]}},
        "shortBox": {"type": "thinsheet", "ior": 1.5, "thickness": 0.4, "sigma_a": [0.3, 0.1, 0.6], "enable_interference": True, "albedo": 1},
        "tallBox": {"type": "diffuse_transmission", "albedo": [0.8, 0.7, 0.5]},
    },
    "zoo_f": {
        "floor": {"type": "oren_nayar", "roughness": dict(_CHECKER, on_color=0.9, off_color=0.05), "albedo": _CHECKER},
        "leftWall": {"type": "phong", "exponent": 6.0, "diffuse_ratio": 0.0, "albedo": [0.63, 0.4, 0.3]},
        "rightWall": dict({"type": "rough_coat", "ior": 1.6, "thickness": 0.5, "sigma_a": 0.0, "distribution": "beckmann", "roughness": 0.35, "albedo": 1,
                           "substrate": dict({"type": "rough_conductor", "distribution": "ggx", "roughness": 0.3, "albedo": 1}, **_CU)}),
        "shortBox": {"type": "thinsheet", "ior": 1.33, "thickness": dict(_CHECKER, on_color=0.8, off_color=0.2), "sigma_a": [0.5, 0.2, 0.1], "albedo": 1},
        "tallBox": {"type": "thinsheet", "ior": 1.7, "albedo": 1},
        "backWall": {"type": "mixed", "ratio": 0.5, "albedo": 1,
                     "bsdf0": {"type": "diffuse_transmission", "albedo": [0.3, 0.6, 0.8]}, "bsdf1": {"type": "lambert", "albedo": 0.9}},
        # (no nested phong / coat / plastic: MixedBsdf, SmoothCoatBsdf, RoughCoatBsdf and TransparencyBsdf::prepareForRender do not forward to the
        # bsdfs inside them and TraceableScene prepares the scene's named bsdfs and the primitives' own only (TraceableScene.hpp:76-84), so an INLINE
        # nested bsdf of the reference evaluates with the uninitialised _invExponent / _brdfFactor / _scaledSigmaA ... its constructor left)
    },
}


def _zoo_d_edit(scene):
    """Sphere primitive + sphere and cube emitters as sampled lights (a16): a glass sphere, an
    emissive sphere and an emissive cube next to the quad light (three lights -> chooseLight's pdf loop)."""
    scene["bsdfs"].append({"name": "glass", "type": "dielectric", "ior": 1.5, "albedo": 1})
    scene["primitives"] += [
        {"name": "ball", "type": "sphere", "bsdf": "glass", "transform": {"position": [0.45, 0.95, 0.35], "scale": 0.28, "rotation": [10, 40, 0]}},
        {"name": "bulb", "type": "sphere", "bsdf": "light", "emission": {"type": "checker", "on_color": [6, 3, 1], "off_color": [1, 3, 6], "res_u": 6, "res_v": 3},
         "transform": {"position": [-0.55, 1.45, 0.1], "scale": 0.12, "rotation": [0, 25, 15]}},
        {"name": "brick", "type": "cube", "bsdf": "light", "emission": [2, 5, 3],
         "transform": {"position": [0.6, 0.12, 0.6], "scale": [0.2, 0.12, 0.16], "rotation": [0, 30, 0]}},
    ]


def cornell_zoo(tmpdir, which, **kw):
    """Cornell box with the named bsdfs replaced (same geometry, same light)."""
    repl = ZOO.get(which, {})

    def edit(scene):
        for i, b in enumerate(scene["bsdfs"]):
            if b["name"] in repl:
                nb = dict(repl[b["name"]])
                nb["name"] = b["name"]
                scene["bsdfs"][i] = nb
    user = kw.pop("edit", None)

    def both(scene):
        edit(scene)
        if which == "zoo_d":
            _zoo_d_edit(scene)
        if user:
            user(scene)
    return variant(CORNELL, str(tmpdir), kw.pop("name", which + ".json"), edit=both, **kw)


def _two_lights(scene):
    """A second, differently coloured quad light low on the left wall: exercises TraceBase::chooseLight (TraceBase.cpp:416-459)."""
    scene["primitives"].append({"name": "light2", "type": "quad", "bsdf": "light", "emission": [3, 9, 14],
                                "transform": {"position": [-0.98, 0.6, 0.2], "scale": [0.3, 0.3, 0.3], "rotation": [0, 0, -90]}})


def _speck_lights(scene):
    """Two more quad emitters a quarter of a millimetre across, next to `light2`: from most of the box their solid angle (1e-8 sr) is
    below the last bit of the 2 pi that Quad::approximateRadiance subtracts four arc cosines from (Quad.cpp:253-281), so the weight
    chooseLight gets for them is rounding noise of either sign -- and a NEGATIVE weight means "unknown": the light receives the mean
    of the known weights (TraceBase.cpp:434-446).  3.3 % of the light choices of this case (264 of 8 000 in the golden) take that branch."""
    _two_lights(scene)
    for i, (pos, rot) in enumerate((([0.4, 1.2, -0.3], [0, 0, 180]), ([-0.97, 1.0, -0.5], [0, 0, -90]))):
        scene["primitives"].append({"name": "speck%d" % i, "type": "quad", "bsdf": "light", "emission": [4e6, 3e6, 2e6],
                                    "transform": {"position": pos, "scale": [2.5e-4, 2.5e-4, 2.5e-4], "rotation": rot}})


def cornell_mesh_light(tmpdir, big=True, **kw):
    """Cornell box whose quad light is replaced by an emissive triangle mesh (sampled: TriangleMesh::sampleDirect,
    TriangleMesh.cpp:411-473): a 168-triangle blob (BVH traversal) or a 2-triangle panel (flat-list traversal)."""
    import numpy as np
    tmpdir = str(tmpdir)
    if big:
        verts, tris = displaced_sphere(8, 12, seed=3)
        verts = verts.copy()
        verts[:, 0:3] = (verts[:, 0:3] - [0, 0.5, 0])*0.5          # radius ~0.22 around the origin
        tris = tris[:, [0, 2, 1, 3]]                               # front faces (the emitting side) outwards
        name = "lamp_blob.wo3"
    else:
        verts = np.array([[-1, 0, -1, 0, -1, 0, 0, 0], [1, 0, -1, 0, -1, 0, 1, 0], [1, 0, 1, 0, -1, 0, 1, 1], [-1, 0, 1, 0, -1, 0, 0, 1]], np.float32)
        tris = np.array([[0, 1, 2, 0], [0, 2, 3, 0]], np.int32)
        name = "lamp_panel.wo3"
    write_wo3(os.path.join(tmpdir, name), verts, tris)
    user = kw.pop("edit", None)

    def edit(scene):
        scene["primitives"] = [p for p in scene["primitives"] if p["name"] != "light"]
        scene["primitives"].append({"name": "lamp", "type": "mesh", "file": name, "smooth": False, "bsdf": "light",
                                    "emission": [9, 7, 4] if big else [17, 12, 4],
                                    "transform": {"position": [0.1, 1.55, 0.0] if big else [-0.005, 1.98, -0.03],
                                                  "scale": [1, 1, 1] if big else [0.235, 1, 0.19]}})
        if user:
            user(scene)
    return variant(CORNELL, tmpdir, kw.pop("name", "mesh_light.json"), edit=edit, **kw)


def _many_cubes(scene):
    """300 small cubes and spheres scattered over the floor: analytic primitives inside a real BVH (not the flat-list path), and
    object/bsdf tables too large for the shading kernels' LDS staging (the global-memory fallback)."""
    import random
    rnd = random.Random(7)
    mats = ["leftWall", "rightWall", "floor", "backWall"]
    for i in range(300):
        x, z = rnd.uniform(-0.9, 0.9), rnd.uniform(-0.9, 0.9)
        sz = rnd.uniform(0.02, 0.05)
        scene["primitives"].append({"name": "c%d" % i, "type": "cube" if i % 3 else "sphere", "bsdf": mats[i % 4],
                                    "transform": {"position": [x, sz, z], "scale": [2*sz, 2*sz, 2*sz] if i % 3 else sz,
                                                  "rotation": [0, rnd.uniform(0, 90), 0]}})


# name -> (builder, kwargs): every per-sample golden under tests/golden/<name>_samples.npz (tools/make_golden.py)
GOLDEN_CASES = {
    "cornell": (cornell, dict(resolution=(48, 27), spp=8)),
    "cornell_nee_off": (cornell, dict(resolution=(32, 18), spp=8, integrator={"enable_light_sampling": False})),
    "cornell_bounce1": (cornell, dict(resolution=(32, 18), spp=8, integrator={"max_bounces": 1})),
    "cornell_bounce2": (cornell, dict(resolution=(32, 18), spp=8, integrator={"max_bounces": 2})),
    "cornell_minb2": (cornell, dict(resolution=(32, 18), spp=8, integrator={"min_bounces": 2})),
    "cornell_onesided": (cornell, dict(resolution=(32, 18), spp=8, integrator={"enable_two_sided_shading": False})),
    "cornell_box_filter": (cornell, dict(resolution=(32, 18), spp=8, edit=lambda s: s["camera"].update(reconstruction_filter="box"))),
    "cornell_two_lights": (cornell, dict(resolution=(32, 18), spp=8, edit=_two_lights)),
    "cornell_speck_lights": (cornell, dict(resolution=(32, 18), spp=8, edit=_speck_lights)),
    "cornell_many_cubes": (cornell, dict(resolution=(32, 18), spp=8, edit=_many_cubes)),
    "cornell_mesh_light": (cornell_mesh_light, dict(resolution=(32, 18), spp=8)),
    "cornell_mesh_light_flat": (lambda t, **kw: cornell_mesh_light(t, big=False, **kw), dict(resolution=(32, 18), spp=8)),
    "cornell_mesh_and_quad_light": (cornell_mesh_light, dict(resolution=(32, 18), spp=8, edit=_two_lights)),
    "zoo_a": (lambda t, **kw: cornell_zoo(t, "zoo_a", **kw), dict(resolution=(48, 27), spp=8)),
    "zoo_b": (lambda t, **kw: cornell_zoo(t, "zoo_b", **kw), dict(resolution=(48, 27), spp=8)),
    "zoo_c": (lambda t, **kw: cornell_zoo(t, "zoo_c", **kw), dict(resolution=(48, 27), spp=8)),
    "zoo_d": (lambda t, **kw: cornell_zoo(t, "zoo_d", **kw), dict(resolution=(48, 27), spp=8)),
    "zoo_e": (lambda t, **kw: cornell_zoo(t, "zoo_e", **kw), dict(resolution=(48, 27), spp=8)),
    "zoo_f": (lambda t, **kw: cornell_zoo(t, "zoo_f", **kw), dict(resolution=(48, 27), spp=8)),
    "materialtest": (materialtest, dict(resolution=(64, 36), spp=4)),
    "materialtest_dielectric": (materialtest, dict(resolution=(48, 27), spp=4, edit=_mt_material({"type": "dielectric", "ior": 1.5, "albedo": 1}))),
    "materialtest_transparency": (materialtest, dict(resolution=(48, 27), spp=4, edit=_mt_material(
        {"type": "transparency", "alpha": 0.5, "albedo": 1, "base": {"type": "lambert", "albedo": [0.8, 0.5, 0.3]}}))),
    "materialtest_rough_dielectric": (materialtest, dict(resolution=(48, 27), spp=4, edit=_mt_material(
        {"type": "rough_dielectric", "ior": 1.5, "distribution": "ggx", "roughness": 0.1, "albedo": 1}))),
}


def write_png(path, img, color_type, depth=8, filters=(0, 1, 2, 3, 4), level=6, palette=None, idat_chunks=1):
    """A PNG (RFC 2083) of `img` ([h, w] or [h, w, channels] integers of `depth` bits) with the given colour type, the scanline filters
    cycling through `filters`, zlib level `level` (0 = stored blocks, 1 = mostly fixed codes, 9 = dynamic codes); test input for the
    host's decoder (tungsten_amd/csrc/host/ImageIO.cpp: loadPng)."""
    import struct
    import zlib
    import numpy as np
    img = np.asarray(img)
    h, w = img.shape[:2]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    a = img.reshape(h, w, ch).astype(np.uint32)
    if depth == 16:
        rows = np.stack([(a >> 8) & 255, a & 255], axis=-1).reshape(h, -1).astype(np.uint8)
    elif depth == 8:
        rows = a.reshape(h, -1).astype(np.uint8)
    else:                                       # 1 / 2 / 4 bits: packed, most significant bits first, rows padded to bytes
        per = 8//depth
        vals = a.reshape(h, -1)
        pad = (-vals.shape[1]) % per
        vals = np.concatenate([vals, np.zeros((h, pad), np.uint32)], axis=1).reshape(h, -1, per)
        rows = sum((vals[:, :, k] << (8 - depth*(k + 1))) for k in range(per)).astype(np.uint8)
    bpp = max(1, ch*depth//8)
    raw = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        f = filters[y % len(filters)]
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if f == 0: out = cur
        elif f == 1: out = cur - left
        elif f == 2: out = cur - prev
        elif f == 3: out = cur - (left + prev)//2
        else:
            pa, pb, pc = np.abs(prev - upleft), np.abs(left - upleft), np.abs(left + prev - 2*upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out = cur - pred
        raw.append(f)
        raw += bytes((out & 255).astype(np.uint8))
        prev = cur
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    z = zlib.compress(bytes(raw), level)
    parts = [z[i*len(z)//idat_chunks:(i + 1)*len(z)//idat_chunks] for i in range(idat_chunks)]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0)))
        if palette is not None:
            f.write(chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1))))
        for p in parts:
            f.write(chunk(b"IDAT", p))
        f.write(chunk(b"IEND", b""))


GOLDEN_CASES["mesh1m"] = (mesh1m, dict(resolution=(48, 27), spp=4))


def tile_terraces(tmpdir, n=24, name="terraces.json", **kw):
    """Axis-aligned geometry on exact grid coordinates -- an n x n floor of unit tiles (two triangles each) at y = 0, a second storey of
    every other tile at y = 2 and walls of unit quads around -- placed 4096 units from the origin: zero-thickness boxes on power-of-two
    planes, the case in which a wide-BVH child's quantised planes coincide with its box (tests of the slack WideBvh.cpp leaves)."""
    import numpy as np
    tmpdir = str(tmpdir)
    verts, tris = [], []

    def quad(p, e0, e1):
        b = len(verts)
        for q in (p, p + e0, p + e0 + e1, p + e1):
            verts.append(list(q) + [0.0, 1.0, 0.0, 0.0, 0.0])
        tris.extend([[b, b + 1, b + 2, 0], [b, b + 2, b + 3, 0]])
    base = np.array([4096.0, 0.0, 4096.0])
    ex, ey, ez = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0])
    for i in range(n):
        for j in range(n):
            quad(base + i*ex + j*ez, ex, ez)
            if (i + j) % 2 == 0:
                quad(base + 2*ey + i*ex + j*ez, ex, ez)
    for i in range(n):
        for k in range(3):
            quad(base + i*ex + k*ey, ex, ey)
            quad(base + i*ez + k*ey, ez, ey)
            quad(base + n*ez + i*ex + k*ey, ex, ey)
            quad(base + n*ex + i*ez + k*ey, ez, ey)
    wo3 = os.path.join(tmpdir, "terraces_%d.wo3" % n)
    write_wo3(wo3, np.array(verts, np.float32), np.array(tris, np.int32))
    scene = {
        "media": [], "bsdfs": [{"name": "grey", "type": "lambert", "albedo": 0.6}],
        "primitives": [{"name": "Terraces", "type": "mesh", "file": os.path.basename(wo3), "smooth": False, "bsdf": "grey", "transform": {}},
                       {"name": "Env", "type": "infinite_sphere", "sample": True, "emission": 1.0}],
        "camera": {"tonemap": "filmic", "resolution": list(kw.get("resolution", (64, 36))), "reconstruction_filter": "tent", "type": "pinhole", "fov": 50,
                   "transform": {"position": [4096 + n/2, 9, 4096 - n], "look_at": [4096 + n/2, 1, 4096 + n/2], "up": [0, 1, 0]}},
        "integrator": {"type": "path_tracer", "min_bounces": 0, "max_bounces": 8, "enable_consistency_checks": False,
                       "enable_two_sided_shading": True, "enable_light_sampling": True},
        "renderer": {"output_file": "", "hdr_output_file": "", "overwrite_output_files": True, "adaptive_sampling": False,
                     "stratified_sampler": False, "scene_bvh": True, "spp": kw.get("spp", 4), "spp_step": kw.get("spp", 4)},
    }
    path = os.path.join(tmpdir, name)
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def terrace_rays(n=24, count=40000, seed=3):
    """Rays at the corners, edges and faces of tile_terraces' tiles: vertical ones through exact grid points and edge midpoints, oblique
    ones from far outside aimed at grid points (origin 1e3 - 1e4 units away: the node-origin / ray-origin cancellation), and rays starting
    ON a tile (distance 0 to its plane)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    g = rs.randint(0, 2*n + 1, size=(count, 2))*0.5                     # grid points and edge midpoints
    target = np.stack([4096.0 + g[:, 0], rs.choice([0.0, 2.0], count), 4096.0 + g[:, 1]], axis=1)
    kind = rs.randint(0, 3, count)
    d = rs.randn(count, 3)
    d[:, 1] = -np.abs(d[:, 1]) - 0.05
    d[kind == 0] = [0.0, -1.0, 0.0]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = np.where(kind == 1, 10.0**rs.uniform(3, 4, count), rs.uniform(1.0, 30.0, count))
    o = target - d*dist[:, None]
    on = kind == 2
    o[on] = target[on] + [0.0, 3.0, 0.0]
    start = on & (rs.rand(count) < 0.5)
    o[start] = target[start]                                             # start on the tile itself
    rays = np.concatenate([o, np.full((count, 1), 1e-4), d, np.full((count, 1), np.inf)], axis=1).astype(np.float32)
    return rays

def cornell_instances(tmpdir, count=40, smooth=(True, False), **kw):
    """Cornell box whose two boxes are replaced by `count` rigid instances of two small master meshes (an `instances`
    primitive, primitives/Instance.cpp): one smooth-shaded rough-conductor blob, one flat-shaded two-material blob; the
    `instances` primitive itself carries a transform too (Instance::prepareForRender composes it, :400-404)."""
    import random
    tmpdir = str(tmpdir)
    for k, (n_lat, n_lon, seed) in enumerate(((7, 10, 11), (5, 8, 12))):
        verts, tris = displaced_sphere(n_lat, n_lon, seed=seed)
        verts = verts.copy()
        verts[:, 1] -= 0.5                                     # centre the master on its origin
        tris = tris[:, [0, 2, 1, 3]].copy()
        if k == 1:
            tris[::2, 3] = 1                                   # every other triangle uses the second bsdf
        write_wo3(os.path.join(tmpdir, "inst_master%d.wo3" % k), verts, tris)
    rnd = random.Random(5)
    inst = []
    for i in range(count):
        inst.append({"id": i % 2, "transform": {"position": [rnd.uniform(-0.8, 0.8), rnd.uniform(0.1, 1.5), rnd.uniform(-0.8, 0.8)],
                                                "rotation": [rnd.uniform(0, 360), rnd.uniform(0, 360), rnd.uniform(0, 360)]}})
    user = kw.pop("edit", None)

    def edit(scene):
        scene["primitives"] = [p for p in scene["primitives"] if p["name"] not in ("shortBox", "tallBox")]
        scene["bsdfs"].append(dict({"name": "copper", "type": "rough_conductor", "distribution": "ggx", "roughness": 0.25, "albedo": 1}, **_CU))
        scene["primitives"].append({
            "name": "swarm", "type": "instances",
            "transform": {"position": [0.05, 0.1, -0.05], "rotation": [0, 20, 0]},
            "masters": [
                {"name": "m0", "type": "mesh", "file": "inst_master0.wo3", "smooth": smooth[0], "bsdf": "copper",
                 "transform": {"scale": 0.3, "rotation": [15, 0, 30]}},
                {"name": "m1", "type": "mesh", "file": "inst_master1.wo3", "smooth": smooth[1], "bsdf": ["leftWall", "rightWall"],
                 "transform": {"scale": [0.22, 0.3, 0.22], "position": [0, 0.02, 0]}}],
            "instances": inst})
        if user:
            user(scene)
    return variant(CORNELL, tmpdir, kw.pop("name", "instances.json"), edit=edit, **kw)


GOLDEN_CASES["cornell_instances"] = (cornell_instances, dict(resolution=(48, 27), spp=8))


def cornell_instance_ties(tmpdir, **kw):
    """Where the `instances` item sits in the reference's top-level order (TraceableScene.hpp:112-134: every finite primitive, the instance set
    being ONE item, in an Embree tree whose visiting order decides equal hits): the Cornell box with its two cubes replaced by three instances of
    a glass box mesh standing ON the floor quad -- identity rotations and y = 0 positions, so the boxes' bottom triangles lie in the floor's
    plane and a ray that went into a box leaves it at a distance the floor reports too, often to the bit.  Which of the two the path takes
    (glass exit or Lambert floor) is decided by that order alone.  A second box leans against the left wall the same way."""
    import numpy as np
    tmpdir = str(tmpdir)
    lo, hi = np.array([-0.25, 0.0, -0.25], np.float32), np.array([0.25, 0.5, 0.25], np.float32)
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], np.float32)   # index = 4 x + 2 y + z
    faces = [((0, 1, 3, 2), (-1, 0, 0)), ((4, 6, 7, 5), (1, 0, 0)), ((0, 4, 5, 1), (0, -1, 0)), ((2, 3, 7, 6), (0, 1, 0)), ((0, 2, 6, 4), (0, 0, -1)), ((1, 5, 7, 3), (0, 0, 1))]
    verts, tris = [], []
    for quad, n in faces:
        base = len(verts)
        for k, c in enumerate(quad):
            verts.append(list(corners[c]) + list(n) + [float(k in (1, 2)), float(k in (2, 3))])
        tris += [[base, base + 1, base + 2, 0], [base, base + 2, base + 3, 0]]
    write_wo3(os.path.join(tmpdir, "tie_box.wo3"), np.array(verts, np.float32), np.array(tris, np.int32))
    user = kw.pop("edit", None)

    def edit(scene):
        scene["primitives"] = [p for p in scene["primitives"] if p["name"] not in ("shortBox", "tallBox")]
        scene["bsdfs"].append({"name": "glass", "type": "dielectric", "ior": 1.5, "albedo": 1})
        scene["primitives"].append({
            "name": "boxes", "type": "instances", "transform": {},
            "masters": [{"name": "box", "type": "mesh", "file": "tie_box.wo3", "smooth": False, "bsdf": "glass", "transform": {}}],
            "instances": [{"id": 0, "transform": {"position": [0.35, 0, 0.3]}}, {"id": 0, "transform": {"position": [-0.3, 0, -0.25]}},
                          {"id": 0, "transform": {"position": [-0.75, 0, 0.4]}}]})      # (x - 0.25 = -1: its side lies in the left wall)
        if user:
            user(scene)
    return variant(CORNELL, tmpdir, kw.pop("name", "instance_ties.json"), edit=edit, **kw)


GOLDEN_CASES["cornell_instance_ties"] = (cornell_instance_ties, dict(resolution=(48, 27), spp=8))

def _thinlens(cateye):
    def edit(scene):
        scene["camera"].update(type="thinlens", focus_distance=6.0, aperture_size=0.12, cateye=cateye)
    return edit


# thin-lens camera with the default disk aperture (cameras/ThinlensCamera.cpp), without and with cat-eye vignetting (which makes
# some direction samples fail -> black samples, PathTracer.cpp:27-28)
GOLDEN_CASES["cornell_thinlens"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_thinlens(0.0)))
GOLDEN_CASES["cornell_thinlens_cateye"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_thinlens(0.35)))
def _thinlens_blade(blades, angle=None):
    def edit(scene):
        ap = {"type": "blade", "blades": blades}
        if angle is not None:
            ap["angle"] = angle
        scene["camera"].update(type="thinlens", focus_distance=6.0, aperture_size=0.12, cateye=0.0, aperture=ap)
    return edit


def cornell_thinlens_bitmap(tmpdir, **kw):
    """Thin-lens camera whose aperture is a bitmap (cameras/ThinlensCamera.cpp:62-63 + BitmapTexture::sample with the MAP_UNIFORM
    distribution): a 24 x 20 grey-scale .png of a ring with a bright notch, written next to the scene."""
    import numpy as np
    tmpdir = str(tmpdir)
    y, x = np.mgrid[0:20, 0:24]
    r = np.hypot((x - 11.5)/11.5, (y - 9.5)/9.5)
    img = np.where((r < 1.0) & (r > 0.45), 90 + 6*x, 0)
    img[2:6, 10:14] = 255
    write_png(os.path.join(tmpdir, "aperture.png"), img, 0, filters=(0, 2), level=6)

    def edit(scene):
        scene["camera"].update(type="thinlens", focus_distance=6.0, aperture_size=0.15, cateye=0.0, aperture="aperture.png")
    return cornell(tmpdir, **dict(kw, edit=edit))


GOLDEN_CASES["cornell_thinlens_bitmap"] = (cornell_thinlens_bitmap, dict(resolution=(48, 27), spp=8))
# n-blade aperture (textures/BladeTexture.cpp): the lens point is a uniform point of one of the polygon's triangles
GOLDEN_CASES["cornell_thinlens_blade5"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_thinlens_blade(5, 0.3)))
GOLDEN_CASES["cornell_thinlens_blade6"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_thinlens_blade(6)))
def cornell_png(tmpdir, **kw):
    """The Cornell box with 8-bit bitmap textures (textures/BitmapTexture.cpp with RGB_LDR texels): a smooth colour pattern on the floor
    (gamma-corrected, interpolated), a raw nearest-neighbour one on the back wall; the .png files are written next to the scene."""
    import numpy as np
    tmpdir = str(tmpdir)
    y, x = np.mgrid[0:48, 0:64]
    img = np.stack([(x*4) % 256, (y*5 + x) % 256, (255 - x*3 - y*2) % 256], axis=-1)
    write_png(os.path.join(tmpdir, "floor_tex.png"), img, 2, filters=(4, 1, 2), level=9)
    write_png(os.path.join(tmpdir, "wall_tex.png"), (img[::4, ::4] // 32)*32 + 16, 2, filters=(0,), level=6)

    def edit(scene):
        scene["bsdfs"].append({"name": "floorTex", "type": "lambert", "albedo": "floor_tex.png"})
        scene["bsdfs"].append({"name": "wallTex", "type": "lambert",
                               "albedo": {"type": "bitmap", "file": "wall_tex.png", "gamma_correct": False, "interpolate": False}})
        for p in scene["primitives"]:
            if p.get("name") == "floor":
                p["bsdf"] = "floorTex"
            if p.get("name") == "backWall":
                p["bsdf"] = "wallTex"
    return cornell(tmpdir, edit=edit, **kw)


GOLDEN_CASES["cornell_png_textures"] = (cornell_png, dict(resolution=(48, 27), spp=8))


def cornell_png_scalar(tmpdir, **kw):
    """Scalar requests of 8-bit bitmaps: a rough conductor whose roughness is the integer channel average of a .png (REQUEST_AVERAGE,
    no gamma), and a cut-out wall whose opacity is the .png's alpha channel (TransparencyBsdf: REQUEST_AUTO)."""
    import numpy as np
    tmpdir = str(tmpdir)
    y, x = np.mgrid[0:32, 0:32]
    rough = np.stack([(x*3 + 20) % 200, (y*4 + 30) % 180, (x + y)*2 % 160], axis=-1) + 10
    write_png(os.path.join(tmpdir, "rough.png"), rough, 2, filters=(2, 4), level=9)
    alpha = np.where(((x//4 + y//4) % 2) == 0, 255, np.where((x + y) % 3 == 0, 128, 0))
    cut = np.stack([x*8 % 256, y*8 % 256, (x*y) % 256, alpha], axis=-1)
    write_png(os.path.join(tmpdir, "cutout.png"), cut, 6, filters=(1, 3), level=6)

    def edit(scene):
        scene["bsdfs"].append(dict({"name": "brushed", "albedo": 1, "type": "rough_conductor", "distribution": "ggx", "roughness": "rough.png"}, **_CU))
        scene["bsdfs"].append({"name": "cutBase", "type": "lambert", "albedo": [0.7, 0.6, 0.2]})
        scene["bsdfs"].append({"name": "cut", "type": "transparency", "base": "cutBase", "alpha": "cutout.png"})
        for p in scene["primitives"]:
            if p.get("name") == "tallBox":
                p["bsdf"] = "brushed"
            if p.get("name") == "shortBox":
                p["bsdf"] = "cut"
    return cornell(tmpdir, edit=edit, **kw)


GOLDEN_CASES["cornell_png_scalar"] = (cornell_png_scalar, dict(resolution=(48, 27), spp=8))


def _thinlens_pivot(scene):
    scene["camera"].update(type="thinlens", focus_distance=1.0, aperture_size=0.12, cateye=0.0, focus_pivot="tallBox")


# focus on a primitive (ThinlensCamera.cpp:206-218): the focus distance is the distance to the origin of the named primitive's frame
GOLDEN_CASES["cornell_thinlens_pivot"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_thinlens_pivot))
GOLDEN_CASES["cornell_thinlens_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_thinlens(0.0), renderer={"stratified_sampler": True}))

def _disks(scene):
    """The quad light becomes a downward disk spot light with a 65 degree emission cone (primitives/Disk.cpp); a checkered,
    non-emissive disk leans against the tall box; a second, small disk light shines sideways from the left wall."""
    scene["primitives"] = [p for p in scene["primitives"] if p["name"] != "light"]
    scene["bsdfs"].append({"name": "checkers", "type": "lambert", "albedo": _CHECKER})
    scene["primitives"] += [
        {"name": "spot", "type": "disk", "bsdf": "light", "emission": [30, 22, 8], "cone_angle": 65,
         "transform": {"position": [-0.005, 1.97, -0.03], "scale": 0.3, "rotation": [180, 0, 0]}},
        {"name": "plate", "type": "disk", "bsdf": "checkers",
         "transform": {"position": [0.35, 0.45, 0.55], "scale": 0.35, "rotation": [60, 25, 10]}},
        {"name": "sidelight", "type": "disk", "bsdf": "light", "emission": [2, 6, 9],
         "transform": {"position": [-0.97, 0.8, 0.3], "scale": 0.15, "rotation": [0, 0, -90]}}]


def _sun_and_sky(scene):
    """Roofless Cornell box under a constant sky (infinite_sphere) and a sun (infinite_sphere_cap, 8 degrees, tilted): two
    infinite lights, both sampled -- chooseLight between them, cap hits on escaping specular/BSDF-sampled paths, the
    "last infinite light that is hit wins" rule of TraceableScene::intersectInfinites."""
    scene["primitives"] = [p for p in scene["primitives"] if p["name"] not in ("light", "ceiling")]
    for i, b in enumerate(scene["bsdfs"]):
        if b["name"] == "tallBox":
            scene["bsdfs"][i] = {"name": "tallBox", "type": "mirror", "albedo": [0.9, 0.9, 0.95]}
    scene["primitives"] += [
        {"name": "sky", "type": "infinite_sphere", "emission": [0.25, 0.35, 0.6], "sample": True},
        {"name": "sun", "type": "infinite_sphere_cap", "emission": [60, 52, 40], "cap_angle": 8, "sample": True,
         "transform": {"rotation": [25, 0, -20]}}]


def _skydome(sample=True, **params):
    """Roofless Cornell box under the procedural sky (primitives/Skydome.cpp: the Hosek-Wilkie model baked into a 512 x 256 image at
    prepareForRender, then an image-based infinite light), the star 35 degrees off the zenith, a mirror box to see it in, and the dimmed
    quad light lying on the floor so that chooseLight weighs two lights."""
    def edit(scene):
        scene["primitives"] = [p for p in scene["primitives"] if p["name"] != "ceiling"]
        for p in scene["primitives"]:
            if p["name"] == "light":
                p["emission"] = [3, 2, 1]
                p["transform"] = {"position": [0.45, 0.01, 0.55], "scale": [0.3, 0.1, 0.2], "rotation": [0, 30, 0]}
        for i, b in enumerate(scene["bsdfs"]):
            if b["name"] == "tallBox":
                scene["bsdfs"][i] = {"name": "tallBox", "type": "mirror", "albedo": [0.9, 0.9, 0.95]}
        scene["primitives"].append(dict({"name": "sky", "type": "skydome", "sample": sample, "transform": {"rotation": [30, 10, -20]}}, **params))
    return edit


GOLDEN_CASES["cornell_skydome"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_skydome()))
GOLDEN_CASES["cornell_skydome_alien"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_skydome(sample=False, temperature=3400.0, turbidity=6.5, intensity=3.0)))


def cornell_bump(tmpdir, **kw):
    """Bump-mapped shading frames (Primitive::setupTangentFrame, primitives/Primitive.cpp:125-163) on every primitive kind with a tangent
    space: a grey-scale .png on the floor quad (glossy, so the frame shows), on the tall cube, on a sphere and on a small smooth mesh; a
    checker bump -- no derivatives, but the frame still comes from the primitive's tangent space -- on the back wall."""
    import numpy as np
    tmpdir = str(tmpdir)
    y, x = np.mgrid[0:40, 0:56]
    img = (127.5 + 90.0*np.sin(x*0.55)*np.cos(y*0.4) + 30.0*np.sin((x + y)*1.3)).clip(0, 255).astype(np.uint8)
    write_png(os.path.join(tmpdir, "bump.png"), img, 0, filters=(0, 1), level=6)
    wo3 = os.path.join(tmpdir, "bump_blob.wo3")
    verts, tris = displaced_sphere(10, 14, seed=5)
    write_wo3(wo3, verts, tris)
    user = kw.pop("edit", None)

    def edit(scene):
        for i, b in enumerate(scene["bsdfs"]):
            if b["name"] == "floor":
                scene["bsdfs"][i] = dict({"name": "floor", "type": "rough_conductor", "distribution": "ggx", "roughness": 0.15, "albedo": [0.8, 0.75, 0.7],
                                          "bump": {"type": "bitmap", "file": "bump.png", "scale": 1.5}}, **_CU)
            elif b["name"] == "tallBox":
                scene["bsdfs"][i] = {"name": "tallBox", "type": "plastic", "ior": 1.5, "albedo": [0.3, 0.5, 0.7], "bump": "bump.png"}
            elif b["name"] == "backWall":
                scene["bsdfs"][i] = {"name": "backWall", "type": "rough_plastic", "ior": 1.4, "distribution": "beckmann", "roughness": 0.2, "albedo": [0.7, 0.7, 0.6],
                                     "bump": {"type": "checker", "on_color": 1.0, "off_color": 0.0, "res_u": 6, "res_v": 6}}
        scene["bsdfs"] += [dict({"name": "ballMat", "type": "rough_conductor", "distribution": "beckmann", "roughness": 0.1, "albedo": 1,
                                 "bump": {"type": "bitmap", "file": "bump.png", "scale": 0.6}}, **_CU),
                           {"name": "blobMat", "type": "rough_dielectric", "ior": 1.5, "distribution": "ggx", "roughness": 0.1, "albedo": 1,
                            "bump": {"type": "bitmap", "file": "bump.png", "scale": 0.5, "interpolate": False}}]
        scene["primitives"] = [p for p in scene["primitives"] if p["name"] != "shortBox"]
        scene["primitives"] += [
            {"name": "ball", "type": "sphere", "bsdf": "ballMat", "transform": {"position": [0.45, 0.3, 0.35], "scale": 0.3, "rotation": [20, 35, 10]}},
            {"name": "blob", "type": "mesh", "file": "bump_blob.wo3", "smooth": True, "bsdf": "blobMat",
             "transform": {"position": [-0.1, 0.35, 0.6], "scale": 0.7, "rotation": [10, 50, 0]}}]
        if user:
            user(scene)
    return cornell(tmpdir, **dict(kw, edit=edit))


GOLDEN_CASES["cornell_bump"] = (cornell_bump, dict(resolution=(48, 27), spp=8))


def _point_lights(scene):
    """Two Dirac point lights (primitives/Point.cpp; one given by emission, one by power) next to the dimmed quad light:
    sampled without random numbers and without MIS, never hit by a ray (TraceBase.cpp:157-158, 281-282, 396-397)."""
    for p in scene["primitives"]:
        if p["name"] == "light":
            p["emission"] = [4, 3, 1]
    scene["primitives"] += [
        {"name": "bulb", "type": "point", "emission": [0.9, 0.5, 0.2], "transform": {"position": [-0.5, 0.4, 0.5]}},
        {"name": "bulb2", "type": "point", "power": [3, 6, 9], "transform": {"position": [0.55, 1.5, 0.3]}}]


def _replace_bsdf(scene, name, bsdf):
    for i, b in enumerate(scene["bsdfs"]):
        if b["name"] == name:
            scene["bsdfs"][i] = dict(bsdf, name=name)


def _prim(scene, name):
    return next(p for p in scene["primitives"] if p["name"] == name)


def _fog(scene):
    """The camera sits in a homogeneous, isotropically scattering medium that no primitive overrides: every segment of every
    path samples a distance (media/HomogeneousMedium.cpp:66-107), volume NEE + phase-function MIS (TraceBase.cpp:323-381)."""
    scene["media"] = scene.get("media", []) + [{"name": "fog", "type": "homogeneous", "sigma_a": [0.02, 0.03, 0.05], "sigma_s": [0.25, 0.22, 0.2]}]
    scene["camera"]["medium"] = "fog"


def _smoke(scene):
    """Media behind boundaries: the tall box becomes a forward-BSDF container of dense, forward-scattering (Henyey-Greenstein)
    smoke, the short box tinted glass (dielectric + absorption-only interior): selectMedium on refraction, shadow rays that
    cross boundaries and pick up transmittance (TraceBase.cpp:62-125), absorption-only distance sampling."""
    scene["media"] = scene.get("media", []) + [
        {"name": "smoke", "type": "homogeneous", "sigma_a": [0.4, 0.5, 0.8], "sigma_s": [3.0, 3.0, 2.5], "density": 1.5,
         "phase_function": {"type": "henyey_greenstein", "g": 0.6}},
        {"name": "tint", "type": "homogeneous", "sigma_a": [2.0, 0.6, 0.3], "sigma_s": 0.0}]
    _replace_bsdf(scene, "tallBox", {"type": "forward", "albedo": 1})
    _replace_bsdf(scene, "shortBox", {"type": "dielectric", "ior": 1.45, "albedo": 1})
    _prim(scene, "tallBox")["int_medium"] = "smoke"
    _prim(scene, "shortBox")["int_medium"] = "tint"


def _fog_and_smoke(scene):
    """Both: the boxes override the camera's fog with their interior and restore it (ext_medium) on the way out."""
    _fog(scene)
    _smoke(scene)
    for n in ("tallBox", "shortBox"):
        _prim(scene, n)["ext_medium"] = "fog"
    scene["integrator"]["max_bounces"] = 12


# participating media: homogeneous media with exponential transmittance
def _rayleigh_fog(scene):
    _fog(scene)
    scene["media"][-1]["phase_function"] = {"type": "rayleigh"}


def volumetric_caustic(tmpdir, **kw):
    """data/example-scenes/volumetric-caustic of the reference (a box filled with a scattering gas behind a forward-BSDF front
    wall, a glass sphere that focuses the light of a null-BSDF emitter into a volumetric caustic), rendered with the path tracer
    instead of its shipped bidirectional integrator (same max_bounces = 6)."""
    user = kw.pop("edit", None)

    def edit(scene):
        keep = {k: scene["integrator"][k] for k in ("min_bounces", "max_bounces", "enable_consistency_checks", "enable_two_sided_shading")}
        scene["integrator"] = dict(keep, type="path_tracer", enable_light_sampling=True)
        if user:
            user(scene)
    return variant(VOLUMETRIC_CAUSTIC, str(tmpdir), kw.pop("name", "volumetric_caustic.json"), edit=edit, **kw)


def non_exponential(tmpdir, gas, **kw):
    """data/example-scenes/non-exponential of the reference: the Cornell box behind a forward-BSDF front wall, filled with a
    scattering gas whose transmittance is not exponential (max_bounces = 3).  The scene defines gas1..gas4 (linear, quadratic,
    double_exponential, pulse) and ships with gas1; `gas` picks one of them, or is a transmittance dict that replaces gas1's."""
    user = kw.pop("edit", None)

    def edit(scene):
        name = gas
        if isinstance(gas, dict):
            scene["media"][0]["transmittance"] = gas
            name = "gas1"
        for p in scene["primitives"]:
            for key in ("int_medium", "ext_medium"):
                if p.get(key) == "gas1":
                    p[key] = name
        if user:
            user(scene)
    return variant(NON_EXPONENTIAL, str(tmpdir), kw.pop("name", "non_exponential.json"), edit=edit, **kw)


for _gas, _tag in (("gas1", "linear"), ("gas2", "quadratic"), ("gas3", "double_exponential"), ("gas4", "pulse"),
                   ({"type": "erlang", "rate": 3.0}, "erlang"), ({"type": "davis", "alpha": 1.6}, "davis")):
    GOLDEN_CASES["non_exponential_" + _tag] = ((lambda g: lambda t, **kw: non_exponential(t, g, **kw))(_gas), dict(resolution=(48, 27), spp=8))
def _area_lights(scene):
    """The scene's emitters are 4.7 x 3.8 mm: Quad::approximateRadiance (Quad.cpp:253-281) subtracts four arc cosines from 2 pi to get
    a solid angle of 1e-5 sr, so the light-selection weights of chooseLight change by several per cent with the last bit of acosf
    (tests/test_gpu_parity.py).  Here they are 40 x 40 cm: the same paths, well-conditioned weights."""
    for p in scene["primitives"]:
        if p.get("bsdf") == "light":
            p["transform"]["scale"] = [0.4, 1.0, 0.4]


# all four gases with emitters large enough for per-sample parity on every implementation of acosf
GOLDEN_CASES["non_exponential_area_lights"] = (lambda t, **kw: non_exponential(t, "gas1", **kw), dict(resolution=(48, 27), spp=8, edit=_area_lights))
GOLDEN_CASES["volumetric_caustic"] = (volumetric_caustic, dict(resolution=(48, 27), spp=8))
def _davis_fog(scene):
    _fog(scene)
    scene["media"][-1]["transmittance"] = {"type": "davis", "alpha": 1.3}


def _davis_weinstein_fog(scene):
    _fog(scene)
    scene["media"][-1]["transmittance"] = {"type": "davis_weinstein", "h": 0.8, "c": 1.2}


GOLDEN_CASES["cornell_fog_davis_weinstein"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_davis_weinstein_fog))
def _interpolated_fog(scene):
    _fog(scene)
    scene["media"][-1]["transmittance"] = {"type": "interpolated", "ratio": 0.35, "tr_a": {"type": "linear", "max_t": 2.0},
                                           "tr_b": {"type": "erlang", "rate": 2.0}}


GOLDEN_CASES["cornell_fog_interpolated"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_interpolated_fog))
GOLDEN_CASES["cornell_fog_davis"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_davis_fog))
GOLDEN_CASES["cornell_fog_rayleigh"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_rayleigh_fog))
def _expfog(scene):
    """media/ExponentialMedium.cpp: the camera's fog thins out upwards (and a little towards the back), density exp(-1.2 (p - unit_point) . dir)."""
    _fog(scene)
    scene["media"][-1].update(type="exponential", falloff_scale=1.2, unit_point=[0.0, 0.3, 0.0], falloff_direction=[0.0, 1.0, 0.2], density=1.6)


def _expfog_and_smoke(scene):
    """The exponential fog outside, homogeneous smoke in the tall box and an ABSORPTION-ONLY exponential medium (a tint that fades with height) in the glass box."""
    _fog_and_smoke(scene)
    for m in scene["media"]:
        if m["name"] == "fog":
            m.update(type="exponential", falloff_scale=0.8, unit_point=[0.0, 1.0, 0.0], falloff_direction=[0.0, -1.0, 0.0])
        if m["name"] == "tint":
            m.update(type="exponential", falloff_scale=2.0, unit_point=[0.3, 0.0, 0.3], falloff_direction=[0.0, 1.0, 0.0])


def _atmosphere(scene):
    """media/AtmosphericMedium.cpp: the camera's fog is a Gaussian ball of haze around the short box (its pivot; density exp(-s^2 (|p - c|^2 - r^2)),
    s = falloff_scale/radius) -- scattering distances through std::erf / exp / Boost's erf_inv in double, optical depths through the A&S erfc."""
    _fog(scene)
    scene["media"][-1].update(type="atmosphere", falloff_scale=1.3, radius=0.7, pivot="shortBox", density=1.4)


def _atmosphere_and_smoke(scene):
    """The haze outside (centre given, not a pivot), homogeneous smoke in the tall box, an ABSORPTION-ONLY atmospheric medium (a tint densest at the
    glass box's heart) inside the short one: AtmosphericMedium::sampleDistance's absorption-only branch and ::transmittance on shadow rays."""
    _fog_and_smoke(scene)
    for m in scene["media"]:
        if m["name"] == "fog":
            m.update(type="atmosphere", falloff_scale=0.9, radius=1.2, center=[0.1, 0.8, -0.2])
        if m["name"] == "tint":
            m.update(type="atmosphere", falloff_scale=1.0, radius=0.3, center=[0.33, 0.3, 0.37])


def _equirectangular(scene):
    """cameras/EquirectangularCamera.cpp: the full sphere seen from a point inside the box (off its axes, rolled a little), longitude across the image."""
    cam = scene["camera"]
    cam["type"] = "equirectangular"
    cam.pop("fov", None)
    cam["transform"] = {"position": [0.15, 0.85, 0.3], "look_at": [-0.4, 0.6, -1.0], "up": [0.1, 1.0, 0.05]}


def _mt_equirectangular(scene):
    """The metric's scene through the equirectangular camera, from where its own camera stands: the BVH scenes' launches (k_camera_rays in front of the wide walk)."""
    scene["camera"]["type"] = "equirectangular"
    scene["camera"].pop("fov", None)


GOLDEN_CASES["materialtest_equirectangular"] = (materialtest, dict(resolution=(64, 32), spp=4, edit=_mt_equirectangular))
def _cubemap(mode):
    """cameras/CubemapCamera.cpp: six 90-degree faces around the same point inside the box; the cross layouts leave pixels outside every face (failed camera samples: black)."""
    def edit(scene):
        _equirectangular(scene)
        scene["camera"]["type"] = "cubemap"
        scene["camera"]["mode"] = mode
    return edit


GOLDEN_CASES["cornell_cubemap_cross"] = (cornell, dict(resolution=(48, 36), spp=8, edit=_cubemap("horizontal_cross")))
GOLDEN_CASES["cornell_cubemap_vertical_sobol"] = (cornell, dict(resolution=(36, 48), spp=8, edit=_cubemap("vertical_cross"), renderer={"stratified_sampler": True}))
GOLDEN_CASES["cornell_cubemap_row"] = (cornell, dict(resolution=(72, 12), spp=8, edit=_cubemap("row")))
GOLDEN_CASES["cornell_equirectangular"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_equirectangular))
GOLDEN_CASES["cornell_equirectangular_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_equirectangular, renderer={"stratified_sampler": True}))
GOLDEN_CASES["cornell_atmosphere"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_atmosphere))
GOLDEN_CASES["cornell_atmosphere_smoke_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_atmosphere_and_smoke, renderer={"stratified_sampler": True}))
GOLDEN_CASES["cornell_expfog"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_expfog))
GOLDEN_CASES["cornell_expfog_smoke_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_expfog_and_smoke, renderer={"stratified_sampler": True}))
GOLDEN_CASES["cornell_fog"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_fog))
GOLDEN_CASES["cornell_smoke"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_smoke))
GOLDEN_CASES["cornell_fog_smoke_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_fog_and_smoke, renderer={"stratified_sampler": True}))
def _cylinders(scene):
    """primitives/Cylinder.cpp: a capped checkered pillar, an open (uncapped) tilted mirror tube one can look into, and an emissive
    tube light next to the dimmed quad light -- a sampled emitter whose approximateRadiance is "unknown" (Cylinder.cpp:280-284),
    so chooseLight gives it the mean of the known weights (TraceBase.cpp:434-446)."""
    for p in scene["primitives"]:
        if p["name"] == "light":
            p["emission"] = [6, 4.5, 1.5]
    scene["primitives"] = [p for p in scene["primitives"] if p["name"] not in ("shortBox", "tallBox")]
    scene["bsdfs"] += [{"name": "checkers", "type": "lambert", "albedo": _CHECKER}, {"name": "tube", "type": "mirror", "albedo": [0.9, 0.85, 0.8]}]
    scene["primitives"] += [
        {"name": "pillar", "type": "cylinder", "bsdf": "checkers", "transform": {"position": [-0.45, 0.5, -0.3], "scale": [0.5, 1.0, 0.5]}},
        {"name": "pipe", "type": "cylinder", "bsdf": "tube", "capped": False,
         "transform": {"position": [0.45, 0.35, 0.3], "scale": [0.5, 0.9, 0.5], "rotation": [70, 25, 0]}},
        {"name": "neon", "type": "cylinder", "bsdf": "light", "emission": [3, 7, 9],
         "transform": {"position": [0.0, 1.6, -0.6], "scale": [0.08, 1.2, 0.08], "rotation": [0, 0, 90]}}]


GOLDEN_CASES["cornell_cylinders"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_cylinders))
GOLDEN_CASES["cornell_point_lights"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_point_lights))
GOLDEN_CASES["cornell_sun_sky"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_sun_and_sky))
GOLDEN_CASES["cornell_disks"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_disks))
GOLDEN_CASES["water_caustic"] = (water_caustic, dict(resolution=(64, 36), spp=4))

# "stratified_sampler": true -- SobolPathSampler dimensions with the tiles' own seeds
_SOBOL = {"stratified_sampler": True}
GOLDEN_CASES["cornell_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, renderer=_SOBOL))
GOLDEN_CASES["zoo_b_sobol"] = (lambda t, **kw: cornell_zoo(t, "zoo_b", **kw), dict(resolution=(48, 27), spp=8, renderer=_SOBOL))
GOLDEN_CASES["materialtest_sobol"] = (materialtest, dict(resolution=(64, 36), spp=4, renderer=_SOBOL))



def _ties(scene):
    """Coincident faces on purpose -- what the reference's top-level Embree tree decides: the light lies IN the ceiling's
    plane; a glass cube sits flush in the back right corner (its faces in the planes of floor, back wall and right wall), a rough-glass slab on
    top of it shares its top face; the tall block (see-through: a thin sheet) stands on the floor; a glass sphere touches the floor in a point;
    a mirror decal lies IN the left wall's plane.  Eleven records, every kind of tie: quad against quad, quad against cube face, cube face
    against cube face, a tangent sphere."""
    scene["bsdfs"] += [{"name": "glass", "type": "dielectric", "ior": 1.5, "albedo": 1},
                       {"name": "frosted", "type": "rough_dielectric", "ior": 1.4, "roughness": 0.2, "distribution": "ggx", "albedo": [0.9, 1.0, 0.9]},
                       {"name": "decal", "type": "mirror", "albedo": [0.9, 0.8, 0.7]}]
    _replace_bsdf(scene, "tallBox", {"type": "thinsheet", "ior": 1.3, "thickness": 0.4, "sigma_a": [0.2, 0.5, 1.0], "albedo": 1})
    _prim(scene, "light")["transform"]["position"] = [-0.005, 2.0, -0.03]
    short = _prim(scene, "shortBox")
    short["transform"] = {"position": [0.7, 0.3, -0.7], "scale": [0.6, 0.6, 0.6]}
    short["bsdf"] = "glass"
    scene["primitives"] += [
        {"name": "slab", "type": "cube", "bsdf": "frosted", "transform": {"position": [0.7, 0.75, -0.7], "scale": [0.6, 0.3, 0.6]}},
        {"name": "ball", "type": "sphere", "bsdf": "glass", "transform": {"position": [-0.55, 0.25, 0.55], "scale": 0.5}},
        {"name": "decal", "type": "quad", "bsdf": "decal", "transform": {"position": [-1, 1, 0.2], "scale": [1, 4, 1], "rotation": [0, 0, 90]}}]
    scene["integrator"]["max_bounces"] = 16


def _round_ties(scene):
    """The same with the round primitives, whose boxes are restated last (Disk::bounds, Cylinder::bounds): a disk light IN the ceiling's plane, a
    capped glass cylinder standing on the floor (its bottom cap in the floor's plane) with a see-through disk lying on its top cap, a mirror disk
    lying IN the floor with a glass sphere standing on it."""
    scene["primitives"] = [p for p in scene["primitives"] if p["name"] not in ("shortBox", "tallBox", "light")]
    scene["bsdfs"] += [{"name": "glass", "type": "dielectric", "ior": 1.5, "albedo": 1},
                       {"name": "sheet", "type": "thinsheet", "ior": 1.3, "thickness": 0.4, "sigma_a": [0.2, 0.5, 1.0], "albedo": 1},
                       {"name": "coaster", "type": "mirror", "albedo": [0.9, 0.8, 0.7]}]
    scene["primitives"] += [
        {"name": "lamp", "type": "disk", "bsdf": "light", "emission": [30, 22, 8], "transform": {"position": [-0.005, 2.0, -0.03], "scale": 0.35, "rotation": [180, 0, 0]}},
        {"name": "pillar", "type": "cylinder", "bsdf": "glass", "transform": {"position": [-0.45, 0.4, -0.3], "scale": [0.5, 0.8, 0.5]}},
        {"name": "lid", "type": "disk", "bsdf": "sheet", "transform": {"position": [-0.45, 0.8, -0.3], "scale": 0.2}},
        {"name": "coaster", "type": "disk", "bsdf": "coaster", "transform": {"position": [0.4, 0.0, 0.4], "scale": 0.35}},
        {"name": "ball", "type": "sphere", "bsdf": "glass", "transform": {"position": [0.4, 0.2, 0.4], "scale": 0.4}}]
    scene["integrator"]["max_bounces"] = 16


GOLDEN_CASES["cornell_ties"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_ties))
GOLDEN_CASES["cornell_round_ties"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_round_ties))
GOLDEN_CASES["cornell_ties_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_ties, renderer=_SOBOL))

# The reference's own PathTraceIntegrator pass loop (`ref_harness integrate`): SampleRecords after every pass + the image.
OUTPUT_TYPES = ("color", "depth", "normal", "albedo", "visibility")


def _outputs(scene):
    scene["primitives"] = [p for p in scene["primitives"] if p["name"] != "ceiling"]
    scene["primitives"].append({"name": "sky", "type": "infinite_sphere", "emission": [0.25, 0.35, 0.6], "sample": True})
    scene["renderer"]["output_buffers"] = [{"type": t, "two_buffer_variance": True, "sample_variance": True} for t in OUTPUT_TYPES]


INTEGRATE_CASES = {
    "cornell_adaptive": (cornell, dict(resolution=(70, 42), spp=72, spp_step=16, renderer={"adaptive_sampling": True})),
    "cornell_adaptive_sobol": (cornell, dict(resolution=(70, 42), spp=64, spp_step=16,
                                             renderer={"adaptive_sampling": True, "stratified_sampler": True})),
    # the renderer block as materialtest.json ships it (materialtest.json:188-190): Sobol' + adaptive, spp_step 16
    "materialtest_as_shipped": (materialtest, dict(resolution=(64, 36), spp=48, spp_step=16,
                                                   renderer={"adaptive_sampling": True, "stratified_sampler": True})),
    # participating media under the adaptive pass loop with the Sobol' sampler
    "cornell_fog_smoke_adaptive": (cornell, dict(resolution=(48, 28), spp=48, spp_step=16, edit=_fog_and_smoke,
                                                 renderer={"adaptive_sampling": True, "stratified_sampler": True})),
}

# renderer.output_buffers: all five outputs with their A/B halves and sample variance (cameras/OutputBuffer.hpp), on a scene with
# specular first vertices (glass, mirror), a textured floor, an emitter in view and a sky for escaping paths; two passes of the
# reference's integrator loop with the Sobol' sampler (adaptive sampling off, so that every pixel takes exactly spp samples)
OUTPUT_CASES = {
    "zoo_a_outputs": (lambda t, **kw: cornell_zoo(t, "zoo_a", **kw), dict(resolution=(48, 28), spp=32, spp_step=16, edit=_outputs,
                                                                       renderer={"adaptive_sampling": False, "stratified_sampler": True})),
}


# ---- the same scenes without coincident faces (oracle-only goldens) ----------------------------------------------------------------
# The Cornell box's two boxes STAND on the floor quad: the bottom face of a see-through box (smoke, glass, cutout) and the floor under it
# are hit at the same distance, and which of the two a ray "sees" is decided by the order the traversal meets them in -- Embree's BVH4 in
# the reference, another tree here.  Lifting every solid by a millimetre removes the tie, and with it EVERY sample in which the oracle
# differs from the reference in these cases (tests/test_oracle_golden.py: bit for bit).
def _lifted(base):
    mk, kw = GOLDEN_CASES[base]

    def make(tmpdir, **kw2):
        path = mk(tmpdir, **kw2)
        with open(path) as f:
            scene = json.load(f)
        for p in scene["primitives"]:
            tr = p.get("transform", {})
            if p["type"] in ("cube", "sphere", "mesh", "cylinder", "disk") and "position" in tr:
                tr["position"][1] += 1e-3
        with open(path, "w") as f:
            json.dump(scene, f)
        return path
    return make, kw


LIFTED_CASES = {base + "_lifted": _lifted(base) for base in ("cornell_smoke", "cornell_fog_smoke_sobol", "cornell_expfog", "cornell_expfog_smoke_sobol", "cornell_fog", "cornell_fog_davis", "cornell_fog_rayleigh",
                                                            "cornell_png_scalar", "zoo_a", "zoo_b", "zoo_e", "zoo_f")}


def _bump_without_the_mesh(scene):
    scene["primitives"] = [p for p in scene["primitives"] if p["name"] != "blob"]


# cornell_bump without its triangle mesh: bump-mapped frames on the quad, the cube, the sphere and the checkered wall alone -- bit-identical;
# what is left of the full case's residual is Embree's division in the mesh's triangle test
LIFTED_CASES["cornell_bump_no_mesh"] = (cornell_bump, dict(resolution=(48, 27), spp=8, edit=_bump_without_the_mesh))


def _crowd(scene):
    """Sixteen records -- the largest flat list -- so that the reference's top-level tree has three levels of nodes: the Cornell box with its
    light in the ceiling's plane and eight more solids on and against the floor, the walls and each other (glass, mirror, see-through), some flush,
    some tangent, one pair interpenetrating."""
    scene["bsdfs"] += [{"name": "glass", "type": "dielectric", "ior": 1.5, "albedo": 1},
                       {"name": "chrome", "type": "mirror", "albedo": [0.9, 0.9, 0.8]},
                       {"name": "sheet", "type": "thinsheet", "ior": 1.3, "thickness": 0.4, "sigma_a": [0.2, 0.5, 1.0], "albedo": 1}]
    _prim(scene, "light")["transform"]["position"] = [-0.005, 2.0, -0.03]
    _replace_bsdf(scene, "tallBox", {"type": "thinsheet", "ior": 1.3, "thickness": 0.4, "sigma_a": [0.2, 0.5, 1.0], "albedo": 1})
    _replace_bsdf(scene, "shortBox", {"type": "dielectric", "ior": 1.45, "albedo": 1})
    add = [("c0", "cube", "glass", [0.8, 0.2, 0.8], [0.4, 0.4, 0.4], None),            # flush in the front right corner of the floor
           ("c1", "cube", "chrome", [0.8, 0.5, 0.8], [0.4, 0.2, 0.4], None),           # on top of it
           ("c2", "cube", "sheet", [-0.85, 0.15, 0.6], [0.3, 0.3, 0.3], [0, 30, 0]),   # rotated, on the floor
           ("s0", "sphere", "glass", [0.0, 0.15, 0.75], 0.3, None),                    # on the floor
           ("s1", "sphere", "chrome", [0.3, 0.15, 0.75], 0.3, None),                   # tangent to s0 ... and the floor
           ("s2", "sphere", "glass", [-0.85, 0.45, 0.6], 0.3, None),                   # on c2
           ("c3", "cube", "glass", [-0.9, 1.0, -0.9], [0.2, 2.0, 0.2], None),          # a column in the back left corner, floor to ceiling
           ("c4", "cube", "sheet", [0.3, 0.3, 0.45], [0.5, 0.5, 0.3], [0, 15, 0])]     # pushed INTO the short box
    for name, typ, bsdf, pos, scale, rot in add:
        tr = {"position": pos, "scale": scale}
        if rot:
            tr["rotation"] = rot
        scene["primitives"].append({"name": name, "type": typ, "bsdf": bsdf, "transform": tr})
    scene["integrator"]["max_bounces"] = 16


GOLDEN_CASES["cornell_crowd"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_crowd))
GOLDEN_CASES["cornell_crowd_sobol"] = (cornell, dict(resolution=(48, 27), spp=8, edit=_crowd, renderer=_SOBOL))
