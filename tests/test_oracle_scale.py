"""The oracle against the reference above golden size (the CPU-side twin of tests/test_gpu_scale.py, which holds the DEVICE to the same hashes):
tests/golden/scale8_<case>.npz / scale64_<case>.npz are 16-bit hashes of the reference's own per-sample radiance at 8 / 64 times the goldens'
samples (tools/make_scale_golden.py).  oracle.c must reproduce every sample bit for bit, except where a scene with a triangle mesh meets a
coincident face (the reference's top-level Embree tree over ALL finite primitives decides those): there the count is pinned at
what round 4's stress renders measured -- exactly, the oracle is deterministic."""
import os
import sys

import numpy as np
import pytest

import oracle_lib
import scenes
import tungsten_amd as tg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import make_scale_golden as msg  # noqa: E402

# oracle samples that are not the reference's (the same scenes as in profiles/r4_oracle_stress_64x_all.txt, which rendered other sample sets); every other case: 0
MEASURED = {("scale8", "cornell_bump"): 5, ("scale64", "cornell_bump"): 46, ("scale8", "mesh1m"): 1, ("scale64", "mesh1m"): 1}


@pytest.mark.parametrize("size,name", [(s, n) for s in ("scale8", "scale64") for n in msg.SIZES[s][2]])
def test_oracle_samples_are_the_references_above_golden_size(size, name, tmp_path):
    if ("materialtest" in name or name == "mesh1m") and not scenes.have_materialtest():
        pytest.skip("materialtest assets (assets/) not present")
    gold = np.load(os.path.join(scenes.GOLDEN, "%s_%s.npz" % (size, name)))
    want, seed = gold["hash"], int(gold["seed"])
    h, w, spp = want.shape
    path, _ = msg.scaled_case(name, str(tmp_path), size)
    flat = tg.FlattenedScene(path)
    tiles = oracle_lib.dice_tiles(w, h, seed)[0] if flat.info.stratified_sampler else None
    got = np.empty((h, w, spp, 3), np.float32)
    for y in range(h):
        for x in range(w):
            ts = None if tiles is None else tiles[(y//16)*((w + 15)//16) + x//16]
            for s in range(spp):
                got[y, x, s] = oracle_lib.trace_sample(flat.desc, seed, x, y, s, tile_seed=ts)
    flat.close()
    differing = int((msg.sample_hash(got) != want).sum())
    assert differing == MEASURED.get((size, name), 0), "%s %s: %d of %d oracle samples are not the reference's (pinned: %d)" % (
        size, name, differing, want.size, MEASURED.get((size, name), 0))
