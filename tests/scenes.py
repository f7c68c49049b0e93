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
