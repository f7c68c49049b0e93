"""The benchmark's workloads (BASELINE.json configs[1..4]): scene descriptions in Tungsten's JSON format, written next to their assets.
Product-side code: bench.py, tools/ and the tests' fixtures (tests/scenes.py re-exports these) build their inputs here.  The Cornell box is
the committed scenes/cornell-box; materialtest is the reference's shipped data/materialtest (meshes, HDRI: data, copied by
__graft_entry__.build() into the git-ignored assets/materialtest); mesh1m and instances10k are generated procedurally with fixed seeds."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "scene.json")
ASSETS = os.path.join(ROOT, "assets")
MATERIALTEST_DIR = os.path.join(ASSETS, "materialtest")


def have_materialtest():
    return os.path.exists(os.path.join(MATERIALTEST_DIR, "materialtest.json"))


def variant(src, dst_dir, name, resolution=None, spp=None, spp_step=None, integrator=None, renderer=None, edit=None):
    """Writes a copy of scene `src` with overrides next to its assets (or into dst_dir for asset-free scenes)."""
    with open(src) as f:
        scene = json.load(f)
    if resolution is not None:
        scene["camera"]["resolution"] = list(resolution)
    if spp is not None:
        scene["renderer"]["spp"] = spp
        scene["renderer"]["spp_step"] = spp_step if spp_step is not None else spp
    elif spp_step is not None:
        scene["renderer"]["spp_step"] = spp_step
    scene["renderer"]["adaptive_sampling"] = False
    scene["renderer"]["stratified_sampler"] = False
    scene["renderer"]["output_file"] = ""
    scene["renderer"]["hdr_output_file"] = ""
    if integrator:
        scene["integrator"].update(integrator)
    if renderer:
        scene["renderer"].update(renderer)
    if edit:
        edit(scene)
    path = os.path.join(dst_dir, name)
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def cornell(tmpdir, **kw):
    return variant(CORNELL, str(tmpdir), kw.pop("name", "cornell.json"), **kw)


def materialtest(tmpdir, **kw):
    """materialtest needs its .wo3/.hdr next to the JSON: link them into tmpdir."""
    for f in os.listdir(MATERIALTEST_DIR):
        if f.endswith(".json"):
            continue
        link = os.path.join(str(tmpdir), f)
        if not os.path.exists(link):
            os.symlink(os.path.join(MATERIALTEST_DIR, f), link)
    return variant(os.path.join(MATERIALTEST_DIR, "materialtest.json"), str(tmpdir), kw.pop("name", "materialtest.json"), **kw)



_CU = {"eta": [0.2004376970, 0.9240334304, 1.1022119527], "k": [3.9129485033, 2.4528477015, 2.1421879552]}
_CHECKER = {"type": "checker", "on_color": [0.725, 0.71, 0.68], "off_color": [0.325, 0.31, 0.25], "res_u": 8, "res_v": 8}


def _mt_material(bsdf):
    """materialtest with the "Material" bsdf swapped (BASELINE.json configs[2]: dielectric variants)."""
    def edit(scene):
        for i, b in enumerate(scene["bsdfs"]):
            if b["name"] == "Material":
                nb = dict(bsdf)
                nb["name"] = "Material"
                scene["bsdfs"][i] = nb
    return edit


# ---- BASELINE.json configs[3]: a procedurally generated ~1M-triangle mesh under an HDRI (or constant) environment ----
def write_wo3(path, verts, tris):
    """MeshIO .wo3 (io/MeshIO.cpp:12-28): u64 numVerts, Vertex{pos3, normal3, uv2} f32, u64 numTris, TriangleI{v0,v1,v2 u32, material i32}."""
    import numpy as np
    with open(path, "wb") as f:
        f.write(np.uint64(len(verts)).tobytes())
        f.write(np.ascontiguousarray(verts, np.float32).tobytes())
        f.write(np.uint64(len(tris)).tobytes())
        f.write(np.ascontiguousarray(tris, np.int32).tobytes())


def displaced_sphere(n_lat=500, n_lon=1000, seed=1):
    """Lat-long sphere of radius ~0.45 displaced by a fixed-seed sum of sines: 2*n_lat*n_lon - 2*n_lon triangles
    (n_lat=500, n_lon=1000 -> 998 000), vertex normals from the analytic gradient direction (unnormalised mix), uv = (u, v)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    k = rs.randint(2, 9, size=(6, 2)).astype(np.float64)
    ph = rs.rand(6)*6.283
    amp = 0.02*rs.rand(6)
    v = (np.arange(n_lat + 1)/n_lat)[:, None]
    u = (np.arange(n_lon)/n_lon)[None, :]
    theta, phi = v*np.pi, u*2*np.pi
    r = 0.45 + sum(amp[i]*np.sin(k[i, 0]*theta + ph[i])*np.cos(k[i, 1]*phi) for i in range(6))
    st, ct = np.sin(theta), np.cos(theta)
    pos = np.stack([r*st*np.cos(phi), r*ct + 0.5, r*st*np.sin(phi)], axis=-1)
    nrm = np.stack([st*np.cos(phi), ct + 0*phi, st*np.sin(phi)], axis=-1)
    uv = np.stack([u + 0*v, v + 0*u], axis=-1)
    verts = np.concatenate([pos, nrm, uv], axis=-1).reshape(-1, 8).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = i*n_lon + j
    b = i*n_lon + (j + 1) % n_lon
    c = (i + 1)*n_lon + j
    d = (i + 1)*n_lon + (j + 1) % n_lon
    t1 = np.stack([a, c, b, 0*a], axis=-1)[1:]            # the first latitude row is degenerate on top ...
    t2 = np.stack([b, c, d, 0*a], axis=-1)[:-1]           # ... and the last at the bottom
    tris = np.concatenate([t1.reshape(-1, 4), t2.reshape(-1, 4)]).astype(np.int32)
    return verts, tris


def mesh1m(tmpdir, resolution=(1920, 1080), spp=512, name="mesh1m.json", n_lat=500, n_lon=1000, **kw):
    """~1M-triangle smooth mesh on a checkered floor, lit by the materialtest HDRI with MIS when its assets are
    present (else a constant environment), rough-conductor material."""
    import json
    tmpdir = str(tmpdir)
    wo3 = os.path.join(tmpdir, "blob_%d_%d.wo3" % (n_lat, n_lon))
    if not os.path.exists(wo3):
        verts, tris = displaced_sphere(n_lat, n_lon)
        write_wo3(wo3, verts, tris)
    env = {"name": "Env", "type": "infinite_sphere", "sample": True, "bsdf": {"albedo": 1, "type": "null"}, "emission": 1.0}
    if have_materialtest():
        link = os.path.join(tmpdir, "envmap.hdr")
        if not os.path.exists(link):
            os.symlink(os.path.join(MATERIALTEST_DIR, "envmap.hdr"), link)
        env["emission"] = "envmap.hdr"
    scene = {
        "media": [],
        "bsdfs": [dict({"name": "metal", "albedo": 1, "type": "rough_conductor", "distribution": "ggx", "roughness": 0.2}, **_CU),
                  {"name": "floor", "type": "lambert", "albedo": dict(_CHECKER, res_u=20, res_v=20)}],
        "primitives": [
            {"name": "Floor", "type": "quad", "bsdf": "floor", "transform": {"position": [0, 0, 0], "scale": [6, 1, 6]}},
            env,
            {"name": "Blob", "type": "mesh", "file": os.path.basename(wo3), "smooth": True, "bsdf": "metal", "transform": {}},
        ],
        "camera": {"tonemap": "filmic", "resolution": list(resolution), "reconstruction_filter": "tent", "type": "pinhole", "fov": 35,
                   "transform": {"position": [1.6, 1.3, 1.9], "look_at": [0, 0.45, 0], "up": [0, 1, 0]}},
        "integrator": {"type": "path_tracer", "min_bounces": 0, "max_bounces": 64, "enable_consistency_checks": False,
                       "enable_two_sided_shading": True, "enable_light_sampling": True},
        "renderer": {"output_file": "", "hdr_output_file": "", "overwrite_output_files": True, "adaptive_sampling": False,
                     "stratified_sampler": False, "scene_bvh": True, "spp": spp, "spp_step": kw.pop("spp_step", spp)},
    }
    path = os.path.join(tmpdir, name)
    with open(path, "w") as f:
        json.dump(scene, f)
    return path



def instances10k(tmpdir, resolution=(1920, 1080), spp=64, name="instances10k.json", count=10000, n_lat=100, n_lon=100, **kw):
    """BASELINE configs[4] in spirit: `count` rigid instances of a ~20 000-triangle master mesh on a jittered grid (seed 1),
    bsdfs cycling lambert / rough_conductor / dielectric / plastic (one master per bsdf: an instance inherits its master's),
    lit by the materialtest HDRI with MIS when available (else a constant environment)."""
    import json
    import random
    tmpdir = str(tmpdir)
    wo3 = os.path.join(tmpdir, "blob_%d_%d.wo3" % (n_lat, n_lon))
    if not os.path.exists(wo3):
        verts, tris = displaced_sphere(n_lat, n_lon)
        verts = verts.copy()
        verts[:, 1] -= 0.5
        write_wo3(wo3, verts, tris)
    env = {"name": "Env", "type": "infinite_sphere", "sample": True, "bsdf": {"albedo": 1, "type": "null"}, "emission": 1.0}
    if have_materialtest():
        link = os.path.join(tmpdir, "envmap.hdr")
        if not os.path.exists(link):
            os.symlink(os.path.join(MATERIALTEST_DIR, "envmap.hdr"), link)
        env["emission"] = "envmap.hdr"
    rnd = random.Random(1)
    side = int(round(count**0.5))
    inst = []
    for i in range(count):
        gx, gz = i % side, i//side
        inst.append({"id": i % 4, "transform": {"position": [(gx - side/2 + rnd.uniform(-0.3, 0.3))*1.1, 0.5 + rnd.uniform(0, 0.4), (gz - side/2 + rnd.uniform(-0.3, 0.3))*1.1],
                                                "rotation": [rnd.uniform(0, 360), rnd.uniform(0, 360), rnd.uniform(0, 360)]}})
    mats = ["diffuse", "metal", "glass", "plastic"]
    scene = {
        "media": [],
        "bsdfs": [{"name": "diffuse", "type": "lambert", "albedo": [0.7, 0.4, 0.3]},
                  dict({"name": "metal", "albedo": 1, "type": "rough_conductor", "distribution": "ggx", "roughness": 0.2}, **_CU),
                  {"name": "glass", "type": "dielectric", "ior": 1.5, "albedo": 1},
                  {"name": "plastic", "type": "plastic", "ior": 1.5, "thickness": 1.0, "sigma_a": [0.2, 0.4, 0.1], "albedo": [0.3, 0.5, 0.7]},
                  {"name": "floor", "type": "lambert", "albedo": dict(_CHECKER, res_u=200, res_v=200)}],
        "primitives": [
            {"name": "Floor", "type": "quad", "bsdf": "floor", "transform": {"position": [0, 0, 0], "scale": [1.2*side, 1, 1.2*side]}},
            env,
            {"name": "Swarm", "type": "instances",
             "masters": [{"name": "m%d" % k, "type": "mesh", "file": os.path.basename(wo3), "smooth": True, "bsdf": mats[k], "transform": {}} for k in range(4)],
             "instances": inst},
        ],
        "camera": {"tonemap": "filmic", "resolution": list(resolution), "reconstruction_filter": "tent", "type": "pinhole", "fov": 40,
                   "transform": {"position": [0.35*side, 0.25*side, 0.55*side], "look_at": [0, 0.5, 0], "up": [0, 1, 0]}},
        "integrator": {"type": "path_tracer", "min_bounces": 0, "max_bounces": 64, "enable_consistency_checks": False,
                       "enable_two_sided_shading": True, "enable_light_sampling": True},
        "renderer": {"output_file": "", "hdr_output_file": "", "overwrite_output_files": True, "adaptive_sampling": False,
                     "stratified_sampler": False, "scene_bvh": True, "spp": spp, "spp_step": kw.pop("spp_step", spp)},
    }
    path = os.path.join(tmpdir, name)
    with open(path, "w") as f:
        json.dump(scene, f)
    return path
