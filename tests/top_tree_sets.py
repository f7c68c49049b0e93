"""Item sets for the top-level-tree tests (tests/test_top_tree.py; tools/make_top_tree_golden.py writes their Embree trees into
tests/golden/top_trees.json)."""
import ctypes as C

import numpy as np

import scenes
import tungsten_amd as tg
from tungsten_amd import capi

# flat lists of quads / cubes / spheres / disks / cylinders among the golden cases, by GOLDEN_CASES name (+ the plain Cornell box and the sphere zoo)
SCENES = ["cornell", "zoo_d", "cornell_smoke", "cornell_fog", "cornell_png_scalar", "cornell_sobol", "zoo_a", "zoo_e", "cornell_disks", "cornell_cylinders", "cornell_ties", "cornell_round_ties", "cornell_crowd"]


# scenes of every other kind -- meshes, mesh emitters, instances, infinite and Dirac lights next to finite primitives: their ITEMS (tgh_scene_items)
# are held to the reference's _finites too, though only flat lists carry the tree so far
ITEM_SCENES = ["cornell_bump", "cornell_mesh_light", "cornell_png_textures", "cornell_instances", "cornell_sun_sky", "cornell_point_lights",
               "materialtest", "mesh1m"]


def _make(name, tmp):
    if name == "cornell":
        return scenes.cornell(tmp, resolution=(16, 9), spp=1)
    if name == "zoo_d":
        return scenes.cornell_zoo(tmp, which="zoo_d", resolution=(16, 9), spp=1)
    mk, kw = scenes.GOLDEN_CASES[name] if name in scenes.GOLDEN_CASES else scenes.LIFTED_CASES[name]
    kw = dict(kw, resolution=(16, 9), spp=1, name=name + "_tt.json")
    if name == "mesh1m":
        kw.update(n_lat=40, n_lon=80)                 # (a small displaced sphere: the item's box is what matters here)
    return mk(tmp, **kw)


def scene_item_boxes(name, tmp):
    """(n, 6) float32: the boxes of the scene's finite primitives in object order -- the items of the reference's user geometry --
    from the library's restatement of Quad / Cube / Sphere / Disk / Cylinder::bounds (tgh_leaf_bounds)."""
    lib = capi.load_library()
    flat = tg.FlattenedScene(_make(name, tmp))
    d = flat.desc.contents
    rec_of = {}
    for r in range(d.num_recs):
        rec_of[d.recs[r].meta & 0x1FFFFFFF] = r
    out = []
    for o in sorted(rec_of):
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        assert lib.tgh_leaf_bounds(C.byref(d.objects[o]), d.recs[rec_of[o]].meta >> 29, lo.ctypes.data, hi.ctypes.data) == 1
        out.append(np.concatenate([lo, hi]))
    flat.close()
    return np.array(out, np.float32)


def random_set(kind, k):
    rs = np.random.RandomState(1000*kind + k)
    n = rs.randint(1, 40) if k % 7 == 0 else rs.randint(1, 17)
    if kind == 0:                                   # random boxes
        c = rs.rand(n, 3)*4 - 2
        h = rs.rand(n, 3)*rs.choice([0.05, 0.5, 2.0])
        b = np.concatenate([c - h, c + h], 1)
    elif kind == 1:                                 # a room: flat quads on grid coordinates, solids inside
        rows = []
        for _ in range(n):
            if rs.rand() < 0.6:
                ax = rs.randint(3)
                lo = np.round(rs.rand(3)*4 - 2, 0)
                hi = lo + np.round(rs.rand(3)*2 + 1, 0)
                hi[ax] = lo[ax]
            else:
                lo = rs.rand(3)*2 - 1
                hi = lo + rs.rand(3)
            rows.append(np.concatenate([lo, hi]))
        b = np.array(rows)
    elif kind == 2:                                 # duplicated boxes: centres coincide, splits fall back to the middle of the list
        m = max(1, n//3)
        c = rs.rand(m, 3)*4 - 2
        h = rs.rand(m, 3)
        base = np.concatenate([c - h, c + h], 1)
        b = base[rs.randint(m, size=n)]
    else:                                           # small-integer coordinates: SAH costs tie
        lo = rs.randint(-3, 3, size=(n, 3)).astype(float)
        b = np.concatenate([lo, lo + rs.randint(0, 3, size=(n, 3))], 1)
    return b.astype(np.float32)
