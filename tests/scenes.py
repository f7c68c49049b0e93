"""Scene fixtures for the tests: variants of the committed Cornell box and, when its assets are
available (oracle/_ref/data populated by __graft_entry__.build() from the reference tree),
of materialtest."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "scene.json")
MATERIALTEST_DIR = os.path.join(ROOT, "oracle", "_ref", "data", "materialtest")
GOLDEN = os.path.join(ROOT, "tests", "golden")


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


# ---- BSDF "zoo": Cornell box variants that exercise every BSDF/texture type on the hot path ----
_CU = {"eta": [0.2004376970, 0.9240334304, 1.1022119527], "k": [3.9129485033, 2.4528477015, 2.1421879552]}
_CHECKER = {"type": "checker", "on_color": [0.725, 0.71, 0.68], "off_color": [0.325, 0.31, 0.25], "res_u": 8, "res_v": 8}

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
}


def cornell_zoo(tmpdir, which, **kw):
    """Cornell box with the named bsdfs replaced (same geometry, same light)."""
    repl = ZOO[which]

    def edit(scene):
        for i, b in enumerate(scene["bsdfs"]):
            if b["name"] in repl:
                nb = dict(repl[b["name"]])
                nb["name"] = b["name"]
                scene["bsdfs"][i] = nb
    user = kw.pop("edit", None)

    def both(scene):
        edit(scene)
        if user:
            user(scene)
    return variant(CORNELL, str(tmpdir), kw.pop("name", which + ".json"), edit=both, **kw)


def _mt_material(bsdf):
    """materialtest with the "Material" bsdf swapped (BASELINE.json configs[2]: dielectric variants)."""
    def edit(scene):
        for i, b in enumerate(scene["bsdfs"]):
            if b["name"] == "Material":
                nb = dict(bsdf)
                nb["name"] = "Material"
                scene["bsdfs"][i] = nb
    return edit


def _two_lights(scene):
    """A second, differently coloured quad light low on the left wall: exercises TraceBase::chooseLight (TraceBase.cpp:416-459)."""
    scene["primitives"].append({"name": "light2", "type": "quad", "bsdf": "light", "emission": [3, 9, 14],
                                "transform": {"position": [-0.98, 0.6, 0.2], "scale": [0.3, 0.3, 0.3], "rotation": [0, 0, -90]}})


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
    "zoo_a": (lambda t, **kw: cornell_zoo(t, "zoo_a", **kw), dict(resolution=(48, 27), spp=8)),
    "zoo_b": (lambda t, **kw: cornell_zoo(t, "zoo_b", **kw), dict(resolution=(48, 27), spp=8)),
    "zoo_c": (lambda t, **kw: cornell_zoo(t, "zoo_c", **kw), dict(resolution=(48, 27), spp=8)),
    "materialtest": (materialtest, dict(resolution=(64, 36), spp=4)),
    "materialtest_dielectric": (materialtest, dict(resolution=(48, 27), spp=4, edit=_mt_material({"type": "dielectric", "ior": 1.5, "albedo": 1}))),
    "materialtest_transparency": (materialtest, dict(resolution=(48, 27), spp=4, edit=_mt_material(
        {"type": "transparency", "alpha": 0.5, "albedo": 1, "base": {"type": "lambert", "albedo": [0.8, 0.5, 0.3]}}))),
    "materialtest_rough_dielectric": (materialtest, dict(resolution=(48, 27), spp=4, edit=_mt_material(
        {"type": "rough_dielectric", "ior": 1.5, "distribution": "ggx", "roughness": 0.1, "albedo": 1}))),
}
