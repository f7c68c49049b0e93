"""What ordering a shading class's slots by hit record would buy (the round-5 review's item 6: "sort live rays by material / closest hit"), measured
instead of argued.  TEST INFRASTRUCTURE (runs on the CPU: the oracle traces the rays).

A workgroup of the wavefront loop owns 4 096 slots; k_shade walks a class's slots in slot order, 64 to a wave, and gathers each hit's 64-byte
attribute record (TgHipTriAttr: two to a 128-byte cache line).  For the slots of one workgroup of the metric's workload -- the work items it is dealt
at the start of a pass -- this script traces the camera rays and two further bounces (directions uniform over the hemisphere the ray came from: where
the hits fall is what matters here, not the BSDF) and counts, per wave of 64 hits, the DISTINCT 128-byte attribute lines the wave touches, in slot order
and with the workgroup's hits sorted by record index.  64 = every lane its own line (nothing to share, sorted or not).

    python tools/sort_locality.py [--scene materialtest|mesh1m] [--workgroups 4]"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import tungsten_amd as tg  # noqa: E402
from tungsten_amd import workloads  # noqa: E402

ITEM_GROUP = 64          # pt_kernels.h: PT_ITEM_GROUP
GRID = 2048              # workgroups of the pool (tghip: grid 2048)
SLOTS = 4096             # slots per workgroup


def lines_per_wave(recs):
    """mean number of distinct 128-byte attribute lines (two 64-byte records each) per wave of 64 consecutive list entries"""
    n = (len(recs)//64)*64
    if n == 0:
        return float("nan")
    lines = (recs[:n] >> 1).reshape(-1, 64)
    return float(np.mean([len(np.unique(w)) for w in lines]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="materialtest")
    ap.add_argument("--workgroups", type=int, default=4)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="tg_sortloc_")
    w, h = (1280, 720) if a.scene == "materialtest" else (1920, 1080)
    path = (workloads.materialtest if a.scene == "materialtest" else workloads.mesh1m)(tmp, resolution=(w, h), spp=4)
    flat = tg.FlattenedScene(path)
    rng = np.random.default_rng(7)
    tiles_x = (w + 15)//16
    pix_slots = tiles_x*((h + 15)//16)*256
    print("%s %dx%d: %d records; one workgroup = %d slots, dealt %d-item groups round-robin over %d workgroups" % (a.scene, w, h, flat.info.num_recs, SLOTS, ITEM_GROUP, GRID))
    print("%-10s %-8s %10s %22s %22s" % ("workgroup", "bounce", "tri hits", "lines/wave slot order", "lines/wave sorted"))
    for b in range(a.workgroups):
        L = np.arange(SLOTS)
        item = ((L//ITEM_GROUP)*GRID + b*(GRID//a.workgroups))*ITEM_GROUP + L % ITEM_GROUP        # (chunk-4 items: item = pixel slot at the start of a pass)
        j = item % pix_slots
        tile, in_tile = j >> 8, j & 255
        px, py = (tile % tiles_x)*16 + (in_tile & 15), (tile//tiles_x)*16 + (in_tile >> 4)
        ok = (px < w) & (py < h)
        rays = np.zeros((SLOTS, 8), np.float32)
        for i in np.nonzero(ok)[0]:
            o, d = oracle_lib.camera_ray(flat.desc, int(px[i]), int(py[i]), float(rng.random()), float(rng.random()))
            rays[i, 0:3], rays[i, 3] = o, 1e-4
            rays[i, 4:7], rays[i, 7] = d, np.inf
        alive = ok.copy()
        for bounce in range(3):
            hits, _, _ = oracle_lib.trace_rays(flat.desc, rays)
            rec = hits["rec"].astype(np.int64)
            hit = alive & (rec >= 0)
            kinds = np.array([flat.rec_kind(int(r)) if r >= 0 else -1 for r in rec]) if hasattr(flat, "rec_kind") else None
            listed = rec[hit]                                   # the class list: the workgroup's hits in slot order
            print("%-10d %-8d %10d %22.1f %22.1f" % (b, bounce, len(listed), lines_per_wave(listed), lines_per_wave(np.sort(listed))))
            # next bounce: from the hit point into the hemisphere the ray came from
            p = rays[:, 0:3] + rays[:, 4:7]*hits["t"][:, None]
            v = rng.normal(size=(SLOTS, 3)).astype(np.float32)
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            flip = (v*rays[:, 4:7]).sum(axis=1) > 0
            v[flip] = -v[flip]
            rays[:, 0:3], rays[:, 4:7] = p, v
            rays[:, 3], rays[:, 7] = 1e-3, np.inf
            alive = hit
            rays[~alive, 7] = -1.0                              # (a ray that can hit nothing)
    flat.close()


if __name__ == "__main__":
    main()
