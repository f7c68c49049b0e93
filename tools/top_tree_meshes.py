"""TEST INFRASTRUCTURE, a PROTOTYPE for the next round (build container only: needs oracle/_ref/ref_harness).  Scenes WITH triangle meshes keep one
tree over all their records; the reference has the mesh as ONE item of its top-level Embree tree.  This tool asks what
walking that tree would buy: it builds the top-level tree over the scene's items (tgh_scene_items: the reference's _finites with their bounds(),
held to the reference's own by tests/test_top_tree.py) with the library's restatement of Embree's builder (tgh_top_tree_build), hands it to the ORACLE with its
`oracle_set_top_items` switch (a leaf = an object; a mesh leaf runs the mesh's own closest-hit query under the hit distance so far), and compares
the oracle's per-sample radiance with the reference's at four times the pixels and twice the samples of the goldens, with and without.

    python tools/top_tree_meshes.py [case ...]        default: the Cornell-box cases with a mesh
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import scenes  # noqa: E402
import tungsten_amd as tg  # noqa: E402
from tungsten_amd import capi  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
SCALE, SPP = int(os.environ.get("STRESS_SCALE", "2")), int(os.environ.get("STRESS_SPP", "16"))     # as tools/oracle_stress.py


def item_tree(flat):
    """(nodes, count): the top-level tree over the scene's items (tgh_scene_items: the reference's _finites with their bounds()), leaves = ~object index."""
    lib = capi.load_library()
    boxes, objs = flat.items()
    boxes = np.ascontiguousarray(boxes, np.float32)
    nodes = (capi.TgHipTopNode*max(len(boxes), 1))()
    count = lib.tgh_top_tree_build(boxes.ctypes.data, len(boxes), nodes, len(nodes))
    for n in range(count):
        for i in range(4):
            if nodes[n].child[i] < 0:
                nodes[n].child[i] = ~int(objs[~nodes[n].child[i]])
    return nodes, count


def main(names):
    for name in names:
        mk, kw = scenes.GOLDEN_CASES[name]
        w0, h0 = kw["resolution"]
        tmp = tempfile.mkdtemp(prefix="tg_items_")
        path = mk(tmp, name=name + ".json", **dict(kw, resolution=(w0*SCALE, h0*SCALE), spp=SPP))
        with open(path) as f:
            sc = json.load(f)
        w, h = sc["camera"]["resolution"]
        spp = sc["renderer"]["spp"]
        out = path + ".bin"
        subprocess.check_call([HARNESS, "samples", path, str(tg.DEFAULT_SEED), str(spp), out], stdout=subprocess.DEVNULL, cwd=tmp)
        ref = np.fromfile(out, np.float32).reshape(h, w, spp, 3)
        flat = tg.FlattenedScene(path)
        nodes, count = item_tree(flat)
        tiles = oracle_lib.dice_tiles(w, h, tg.DEFAULT_SEED)[0] if flat.info.stratified_sampler else None
        res = []
        for items in (False, True):
            desc = flat.desc
            if items:
                if not count:
                    res.append(None)
                    continue
                d2 = tg.TgHipSceneDesc.from_buffer_copy(flat.desc.contents)
                d2.top_nodes = C.cast(nodes, C.POINTER(capi.TgHipTopNode))
                d2.num_top_nodes = count
                desc = C.pointer(d2)
            oracle_lib._lib.oracle_set_top_items(1 if items else 0)
            try:
                differing = 0
                for y in range(h):
                    for x in range(w):
                        ts = None if tiles is None else tiles[(y//16)*((w + 15)//16) + x//16]
                        for s in range(spp):
                            g = np.asarray(oracle_lib.trace_sample(desc, tg.DEFAULT_SEED, x, y, s, tile_seed=ts))
                            differing += not (g == ref[y, x, s]).all()
            finally:
                oracle_lib._lib.oracle_set_top_items(0)
            res.append(differing)
        flat.close()
        print("%-32s %7d samples: %4d differ as built, %s with the top-level tree over items (%d nodes)" % (
            name, h*w*spp, res[0], "n/a" if res[1] is None else "%4d" % res[1], count), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["cornell_bump", "cornell_mesh_light", "cornell_mesh_light_flat", "cornell_png_textures", "cornell_instances", "materialtest", "mesh1m"])
