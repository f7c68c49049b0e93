"""TEST INFRASTRUCTURE, a PROTOTYPE for the next round (build container only: needs oracle/_ref/ref_harness).  Scenes WITH triangle meshes keep one
tree over all their records; the reference has the mesh as ONE item of its top-level Embree tree.  This tool asks what
walking that tree would buy: it builds the top-level tree over the scene's finite primitives -- analytic ones by their restated bounds(), a mesh by
the box of its triangles' vertices -- with the library's restatement of Embree's builder (tgh_top_tree_build), hands it to the ORACLE with its
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


def item_tree(desc):
    """(nodes, count): the top-level tree over the objects that have records, leaves = ~object index; None when an object's box is unknown."""
    lib = capi.load_library()
    d = desc.contents
    recs = np.ctypeslib.as_array(C.cast(d.recs, C.POINTER(C.c_float)), shape=(d.num_recs, 12))
    meta = recs.view(np.uint32)[:, 3]
    kind, obj = meta >> 29, meta & 0x1FFFFFFF
    boxes, objs = [], []
    for o in sorted(set(obj.tolist())):
        sel = obj == o
        k = int(kind[sel][0])
        if k == 0:                                   # triangles: v0, v0 + e1, v0 + e2 (an ulp from the mesh's own vertices at most)
            a, b, c = recs[sel, 0:3], recs[sel, 4:7], recs[sel, 8:11]
            v = np.concatenate([a, a + b, a + c])
            boxes.append(np.concatenate([v.min(axis=0), v.max(axis=0)]))
        else:
            lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
            if lib.tgh_leaf_bounds(C.byref(d.objects[o]), k, lo.ctypes.data, hi.ctypes.data) != 1:
                return None, 0
            boxes.append(np.concatenate([lo, hi]))
        objs.append(o)
    boxes = np.ascontiguousarray(boxes, np.float32)
    nodes = (capi.TgHipTopNode*max(len(boxes), 1))()
    count = lib.tgh_top_tree_build(boxes.ctypes.data, len(boxes), nodes, len(nodes))
    for n in range(count):
        for i in range(4):
            if nodes[n].child[i] < 0:
                nodes[n].child[i] = ~objs[~nodes[n].child[i]]
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
        nodes, count = item_tree(flat.desc)
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
    main(sys.argv[1:] or ["cornell_bump", "cornell_mesh_light", "cornell_mesh_light_flat", "cornell_png_textures", "materialtest", "mesh1m"])
