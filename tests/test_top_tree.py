"""The reference's top-level Embree tree, restated (csrc/host/EmbreeTopTree.cpp; include/tungsten_hip.h: TgHipTopNode): built from the items'
boxes by tgh_top_tree_build it must be the BVH4 Embree 2.11 builds over the same user-geometry items -- same children in the same slots with the
same boxes --, because where faces coincide the order in which a ray visits that tree decides what it hits.  tests/golden/top_trees.json holds
trees read out of the reference's own Embree (oracle/ref_embree_tree.cpp, tools/make_top_tree_golden.py) for the flat-list golden scenes and
240 seeded random item sets; in the build container the same comparison runs live on fresh random sets."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import scenes
import top_tree_sets
import tungsten_amd as tg
from tungsten_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_embree_tree")


def tree_text(nodes, count):
    """The tree in ref_embree_tree's notation."""
    def hexf(v):
        return "".join("%08x " % x for x in np.asarray(v, np.float32).view(np.uint32))

    def dump(c):
        if c < 0:
            return "L%d " % (~c)
        n = nodes[c]
        s = "N( "
        for i in range(4):
            if n.child[i] == capi.TGHIP_TOP_EMPTY:
                s += "- "
                continue
            s += hexf(list(n.lower[i])) + hexf(list(n.upper[i])) + dump(n.child[i])
        return s + ") "
    return (dump(0) if count else "L0 ").strip()


def build(boxes):
    lib = capi.load_library()
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    nodes = (capi.TgHipTopNode*max(len(boxes), 1))()
    count = lib.tgh_top_tree_build(boxes.ctypes.data, len(boxes), nodes, len(nodes))
    assert 0 <= count < max(len(boxes), 2)
    return nodes, count


def test_restated_builder_builds_embrees_trees():
    with open(os.path.join(scenes.GOLDEN, "top_trees.json")) as f:
        gold = json.load(f)
    assert len(gold) >= 200
    kinds = set()
    for g in gold:
        boxes = np.array([[int(v, 16) for v in row] for row in g["boxes"]], np.uint32).view(np.float32)
        nodes, count = build(boxes)
        assert tree_text(nodes, count) == g["tree"], g["name"]
        kinds.add(g["name"].split(":")[0] + ":" + g["name"].split(":")[1] if g["name"].startswith("random") else "scene")
    assert kinds == {"scene", "random:0", "random:1", "random:2", "random:3"}


@pytest.mark.parametrize("name", top_tree_sets.SCENES)
def test_scene_descriptions_carry_the_tree_of_their_items(name, tmp_path):
    """TgHipSceneDesc::top_nodes of a flat list of analytic primitives (quads, cubes, spheres, disks, cylinders) is the tree over the scene's objects in object order (the reference's
    _finites), its leaves naming the objects' records; the fixtures hold Embree's tree over those very boxes."""
    with open(os.path.join(scenes.GOLDEN, "top_trees.json")) as f:
        gold = {g["name"]: g for g in json.load(f)}
    g = gold["scene:" + name]
    boxes = top_tree_sets.scene_item_boxes(name, tmp_path)
    assert [["%08x" % v for v in row] for row in boxes.view(np.uint32).tolist()] == g["boxes"]
    flat = tg.FlattenedScene(top_tree_sets._make(name, tmp_path))
    d = flat.desc.contents
    assert 0 < d.num_top_nodes < d.num_recs
    item_of_rec = {}
    objs = sorted(set(d.recs[r].meta & 0x1FFFFFFF for r in range(d.num_recs)))
    for r in range(d.num_recs):
        item_of_rec[r] = objs.index(d.recs[r].meta & 0x1FFFFFFF)
    text = tree_text(d.top_nodes, d.num_top_nodes)
    # leaves name records there, items in the fixture
    import re
    text = re.sub(r"L(\d+) ", lambda m: "L%d " % item_of_rec[int(m.group(1))], text + " ").strip()
    flat.close()
    assert text == g["tree"]


@pytest.mark.parametrize("name", top_tree_sets.SCENES)
def test_item_boxes_are_the_references_bounds(name, tmp_path):
    """Quad / Cube / Sphere / Disk / Cylinder::bounds as the library restates them from the flattened objects (tgh_leaf_bounds), in object order,
    against the reference's own bounds() of the scene's finite primitives in scene order (oracle/ref_harness.cpp: bounds), bit for bit --
    the boxes AND the order of the items of the reference's top-level user geometry."""
    with open(os.path.join(scenes.GOLDEN, "prim_bounds.json")) as f:
        gold = json.load(f)[name]
    boxes = top_tree_sets.scene_item_boxes(name, tmp_path)
    assert [["%08x" % v for v in row] for row in boxes.view(np.uint32).tolist()] == gold


@pytest.mark.parametrize("name", top_tree_sets.SCENES + top_tree_sets.ITEM_SCENES)
def test_scene_items_are_the_references_finites(name, tmp_path):
    """tgh_scene_items: the finite primitives of a scene in scene order with their bounds() as csrc/host/Scene.cpp restates them per primitive class
    -- quads, cubes, spheres, disks, cylinders, triangle meshes (static, emissive, bump-mapped), `instances` -- against the reference's own bounds()
    of its _finites (oracle/ref_harness.cpp: bounds), bit for bit and in order: the items the reference's top-level Embree tree is built over, for
    every kind of scene (infinite and Dirac emitters are not among them)."""
    if name in ("materialtest", "mesh1m") and not scenes.have_materialtest():
        pytest.skip("materialtest assets (assets/) not present")
    with open(os.path.join(scenes.GOLDEN, "prim_bounds.json")) as f:
        gold = json.load(f)[name]
    flat = tg.FlattenedScene(top_tree_sets._make(name, tmp_path))
    boxes, objects = flat.items()
    d = flat.desc.contents
    assert [["%08x" % v for v in row] for row in boxes.view(np.uint32).tolist()] == gold
    assert (np.diff(objects) > 0).all() and 0 <= objects[0] and objects[-1] < d.num_objects
    rec_objects = set(d.recs[r].meta & 0x1FFFFFFF for r in range(min(d.num_recs, 4096)))
    assert rec_objects <= set(objects.tolist()) or d.num_instances          # every record belongs to an item (masters of instances aside)
    flat.close()


def test_scenes_that_are_not_such_lists_carry_no_tree(tmp_path):
    for mk, kw in (scenes.GOLDEN_CASES["cornell_bump"], scenes.GOLDEN_CASES["cornell_instances"], scenes.GOLDEN_CASES["cornell_mesh_light"]):
        flat = tg.FlattenedScene(mk(tmp_path, **dict(kw, resolution=(16, 9), spp=1)))
        assert flat.desc.contents.num_top_nodes == 0 and not flat.desc.contents.top_nodes
        flat.close()
    # fewer than two items: Embree's root is the leaf itself; invalid boxes: Embree drops the item, the builder refuses
    assert build(np.array([[0, 0, 0, 1, 1, 1]], np.float32))[1] == 0
    assert build(np.array([[0, 0, 0, 1, 1, 1], [2, 0, 0, 1, 1, 1]], np.float32))[1] == 0
    assert build(np.array([[0, 0, 0, 1, 1, 1], [0, 0, 0, np.inf, 1, 1]], np.float32))[1] == 0
    # ... and builds the tree over the others, which keep their ids (Embree's primref pass skips an invalid item): three items, the middle one
    # invalid -> one node whose two leaves are items 0 and 2
    nodes3, count3 = build(np.array([[0, 0, 0, 1, 1, 1], [0, 0, 0, np.nan, 1, 1], [4, 0, 0, 5, 1, 1]], np.float32))
    assert count3 == 1 and sorted(~c for c in nodes3[0].child if c < 0) == [0, 2]
    # renderer.scene_bvh = false: the reference commits no Embree scene and asks its primitives in scene order (TraceableScene.hpp:175-181)
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, name="nobvh.json", resolution=(16, 9), spp=1, renderer={"scene_bvh": False}))
    assert flat.desc.contents.num_top_nodes == 0 and not flat.desc.contents.top_nodes
    flat.close()
    flat = tg.FlattenedScene(scenes.cornell(tmp_path, name="bvh.json", resolution=(16, 9), spp=1))
    assert flat.desc.contents.num_top_nodes > 0
    flat.close()
    lib = capi.load_library()
    nodes = (capi.TgHipTopNode*1)()
    b = np.array([[0, 0, 0, 1, 1, 1], [2, 0, 0, 3, 1, 1], [4, 0, 0, 5, 1, 1], [6, 0, 0, 7, 1, 1], [8, 0, 0, 9, 1, 1], [10, 0, 0, 11, 1, 1]], np.float32)
    assert lib.tgh_top_tree_build(b.ctypes.data, len(b), nodes, 1) == -1                  # capacity


@pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref/ref_embree_tree (the reference's Embree) not built")
def test_restated_builder_against_the_references_embree_live():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_top_tree_golden as mk
    sets = [top_tree_sets.random_set(kind, 5000 + k) for kind in range(4) for k in range(250)]
    # ... and large sets, past the sizes at which Embree's builder spawns tasks (256 items) and bins / partitions in parallel (3072)
    rs = np.random.RandomState(9)
    for n in (300, 3100):
        c, h = rs.rand(n, 3)*10 - 5, rs.rand(n, 3)*0.3
        sets.append(np.concatenate([c - h, c + h], 1).astype(np.float32))
        lo = rs.randint(-6, 6, size=(n, 3)).astype(np.float32)
        sets.append(np.concatenate([lo, lo + rs.randint(0, 3, size=(n, 3))], 1).astype(np.float32))
    import sys as _sys
    _sys.setrecursionlimit(20000)
    trees = mk.embree_trees(sets)
    for s, t in zip(sets, trees):
        nodes, count = build(s)
        assert tree_text(nodes, count) == t.strip()
