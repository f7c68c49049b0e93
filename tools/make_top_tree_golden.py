"""TEST INFRASTRUCTURE (build container only: needs oracle/_ref/ref_embree_tree, built by `make -f oracle/Makefile.ref` from the reference's
vendored Embree, and oracle/_ref/ref_harness).  Writes tests/golden/prim_bounds.json -- the reference's own bounds() of the finite primitives of the
flat-list golden scenes, in scene order -- and tests/golden/top_trees.json: item boxes and, for each set, the BVH4 the reference's own Embree built over them
(oracle/ref_embree_tree.cpp reads it out of Embree's structures) -- what csrc/host/EmbreeTopTree.cpp restates and tests/test_top_tree.py
compares.  Sets: the leaf boxes of the flat-list golden scenes in object order, and seeded random sets of four kinds (random boxes; rooms of
flat quads on grid coordinates with solids inside; duplicated boxes; small-integer coordinates, where SAH costs tie).

    python tools/make_top_tree_golden.py [sets-per-kind]
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import top_tree_sets  # noqa: E402

TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_embree_tree")


def embree_trees(sets):
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "b.bin"), "wb") as f:
            f.write(struct.pack("<I", len(sets)))
            for s in sets:
                f.write(struct.pack("<I", len(s)))
                f.write(np.ascontiguousarray(s, np.float32).tobytes())
        subprocess.check_call([TOOL, os.path.join(tmp, "b.bin"), os.path.join(tmp, "o.txt")])
        with open(os.path.join(tmp, "o.txt")) as f:
            return f.read().splitlines()


if __name__ == "__main__":
    per_kind = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    names, sets = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for name in top_tree_sets.SCENES:
            b = top_tree_sets.scene_item_boxes(name, tmp)
            names.append("scene:" + name)
            sets.append(b)
    for kind in range(4):
        for k in range(per_kind):
            names.append("random:%d:%d" % (kind, k))
            sets.append(top_tree_sets.random_set(kind, k))
    # the reference's own bounds() of the scenes' finite primitives (oracle/ref_harness.cpp: bounds), which the scene sets above must equal
    HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    bounds = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in top_tree_sets.SCENES + top_tree_sets.ITEM_SCENES:
            path = top_tree_sets._make(name, tmp)
            subprocess.check_call([HARNESS, "bounds", path, os.path.join(tmp, "b.txt")], stdout=subprocess.DEVNULL, cwd=os.path.dirname(path))
            with open(os.path.join(tmp, "b.txt")) as f:
                bounds[name] = [line.split() for line in f.read().splitlines()]
    with open(os.path.join(scenes.GOLDEN, "prim_bounds.json"), "w") as f:
        json.dump(bounds, f, separators=(",", ":"))
    trees = embree_trees(sets)
    assert len(trees) == len(sets)
    out = [{"name": n, "boxes": [["%08x" % v for v in row] for row in np.ascontiguousarray(s, np.float32).view(np.uint32).tolist()], "tree": t.strip()}
           for n, s, t in zip(names, sets, trees)]
    with open(os.path.join(scenes.GOLDEN, "top_trees.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("%d sets, %d bytes" % (len(out), os.path.getsize(os.path.join(scenes.GOLDEN, "top_trees.json"))))
