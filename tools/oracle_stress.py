"""TEST INFRASTRUCTURE (build container only: needs oracle/_ref/ref_harness).  The golden cases in which the oracle is the reference bit for
bit (tests/test_oracle_golden.py: BIT_IDENTICAL, and the LIFTED twins) at four times the pixels and twice the samples of the committed goldens:
the reference's own per-sample radiance from the harness against oracle.c, float32 ==.

    python tools/oracle_stress.py [case ...]          LIFT=1e-3 python tools/oracle_stress.py case    (every solid lifted off the floor)

End of round 4, with the reference's top-level Embree tree restated: 73 of the 75 cases and twins have no differing
sample in 36 864-147 456 (profiles/r4_oracle_stress_top_tree.txt); cornell_bump 5 -- the tall box's bottom face against the floor quad in a
scene that, having a mesh, carries no top-level tree here --, mesh1m 1.  (Before the tree: box-bottom ties in the fog cases, the seam of floor
and wall under the Sobol' sampler's diagonal points, and -- until Embree's leaf test was restated -- one-ulp differences where Embree culls the
flat box of the light a multi-segment shadow ray ends on.)"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import scenes  # noqa: E402
import test_oracle_golden as T  # noqa: E402
import tungsten_amd as tg  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
LIFT = os.environ.get("LIFT")
SCALE, SPP = int(os.environ.get("STRESS_SCALE", "2")), int(os.environ.get("STRESS_SPP", "16"))     # STRESS_SCALE=4 STRESS_SPP=32: 64 times the goldens' samples
names = sys.argv[1:] or sorted(T.BIT_IDENTICAL) + sorted(scenes.LIFTED_CASES)
for name in names:
    mk, kw = scenes.GOLDEN_CASES[name] if name in scenes.GOLDEN_CASES else scenes.LIFTED_CASES[name]
    w0, h0 = kw["resolution"]
    tmp = tempfile.mkdtemp(prefix="tg_stress_")
    path = mk(tmp, name=name + ".json", **dict(kw, resolution=(w0*SCALE, h0*SCALE), spp=SPP))
    with open(path) as f:
        sc = json.load(f)
    if LIFT:
        for p in sc["primitives"]:
            tr = p.get("transform", {})
            if p["type"] in ("cube", "sphere", "mesh", "cylinder", "disk") and "position" in tr:
                tr["position"][1] += float(LIFT)
        with open(path, "w") as f:
            json.dump(sc, f)
    w, h = sc["camera"]["resolution"]
    spp = sc["renderer"]["spp"]
    out = path + ".bin"
    subprocess.check_call([HARNESS, "samples", path, str(tg.DEFAULT_SEED), str(spp), out], stdout=subprocess.DEVNULL, cwd=os.path.dirname(path))
    ref = np.fromfile(out, np.float32).reshape(h, w, spp, 3)
    flat = tg.FlattenedScene(path)
    tiles = oracle_lib.dice_tiles(w, h, tg.DEFAULT_SEED)[0] if flat.info.stratified_sampler else None
    t0, differing, beyond, first = time.time(), 0, 0, None
    for y in range(h):
        for x in range(w):
            ts = None if tiles is None else tiles[(y//16)*((w + 15)//16) + x//16]
            for s in range(spp):
                g = np.asarray(oracle_lib.trace_sample(flat.desc, tg.DEFAULT_SEED, x, y, s, tile_seed=ts))
                r = ref[y, x, s]
                if not (g == r).all():
                    differing += 1
                    beyond += bool(np.abs(g - r).max() > 1e-3*(np.abs(r).max() + 1e-3))
                    first = first or (x, y, s)
    flat.close()
    print("%-40s %6d samples: %4d not bit-identical (%d beyond 1e-3)%s  %.0f s" % (name, h*w*spp, differing, beyond, "  first at %r" % (first,) if first else "", time.time() - t0), flush=True)
