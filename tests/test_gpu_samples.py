"""Per-sample parity of the HIP path: the radiance of every individual (pixel, sample) -- what PathTracer::traceSample
returns (integrators/path_tracer/PathTracer.cpp:14-149) -- read back through TGHIP_PASS_SAMPLES / tghip_download_samples and
compared with the reference's own per-sample output (tests/golden/*_samples.npz, rendered by oracle/ref_harness.cpp with the
shared counter-based random stream) AND with the oracle's, by the metric tests/test_oracle_golden.py uses: a sample agrees when every
channel is within 1e-3 of the other's (relative to the sample's largest channel).

Measured at the end of round 4 (profiles/r4_device_diverge_top_tree.jsonl), with every libm function the path calls, Embree's triangle
arithmetic and the reference's top-level Embree tree (builder and walk) restated on the device: in NONE of the 549 504 samples of the 62 cases (nor of the 51 840 of the five tie scenes added after the table)
does the device leave the oracle's path or the reference's, and its float32 radiance is BOTH's bit for bit in every sample.  No case is
listed in test_oracle_golden.DIVERGING any more (a case listed there would be held to 1.5 x its measured count + 5 samples), so every case
asserts bit-equality with the oracle and with the reference."""
import json
import os

import numpy as np
import pytest

import oracle_lib
import scenes
import tungsten_amd as tg
from test_oracle_golden import DIVERGING, PINNED, diverge_bound, _oracle_samples

pytestmark = pytest.mark.gpu
SEED = tg.DEFAULT_SEED
TABLE = os.environ.get("TG_DIVERGE_TABLE")     # when set: append one JSON line per case

# device samples allowed to leave the ORACLE's path: measured 0 in every case
DEVICE_VS_ORACLE = 5
# Cases whose device radiance is NOT the oracle's bit for bit although every sample is on the oracle's path: none.  (Until round 4 the shadow
# kernel multiplied a stored f*e/pdf*mis by the transmittance it found, where the reference multiplies e by it FIRST (TraceBase.cpp:144-174,
# 246-285): 19 cases with media, see-through surfaces or mesh emitters differed in the last bit of up to 70 % of their samples.  Those scenes'
# shading kernels now leave the factors apart -- PathState::nee_factors, A_NEE0 .. A_NEE2 -- and k_trace_shadow<., FORWARD> multiplies in the
# reference's order.)
ULP_LEVEL = set()


def _skip(name):
    if name == "water_caustic" and not scenes.have_water_caustic():
        pytest.skip("water-caustic assets (assets/) not present")
    if ("materialtest" in name or name == "mesh1m") and not scenes.have_materialtest():
        pytest.skip("materialtest assets (assets/) not present")


def diverging(got, ref):
    err = np.abs(got - ref).max(axis=-1)
    return err > 1e-3*(np.abs(ref).max(axis=-1) + 1e-3)


@pytest.mark.parametrize("name", sorted(scenes.GOLDEN_CASES))
def test_device_samples_match_the_reference_per_sample(name, tmp_path):
    _skip(name)
    mk, kw = scenes.GOLDEN_CASES[name]
    gold = np.load(os.path.join(scenes.GOLDEN, name + "_samples.npz"))
    ref = gold["samples"]
    seed = int(gold["seed"])
    h, w, spp, _ = ref.shape
    path = mk(tmp_path, name=name + ".json", **kw)
    r = tg.Renderer(path, seed=seed)
    assert (r.width, r.height) == (w, h)
    sobol = bool(r.info.stratified_sampler)
    got = r.trace_samples(0, spp, seed=seed, tile_seeds=oracle_lib.dice_tiles(w, h, seed)[0] if sobol else None)
    # the framebuffer of the same pass is the per-pixel sum of those samples (NaN/Inf samples aside, which the goldens do not hold)
    mean, ssum, count = r.image()
    r.close()
    assert (count == spp).all()
    assert np.allclose(got.sum(axis=2), ssum, rtol=1e-5, atol=1e-6)
    ora = _oracle_samples(mk, kw, name, tmp_path, ref, seed)
    off_ref, off_oracle = diverging(got, ref), diverging(got, ora)
    bit_ref = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=-1)
    bit_oracle = (got.view(np.uint32) == ora.view(np.uint32)).all(axis=-1)
    if TABLE:
        with open(TABLE, "a") as f:
            f.write(json.dumps({"case": name, "samples": int(off_ref.size), "device_vs_ref": int(off_ref.sum()), "device_vs_oracle": int(off_oracle.sum()),
                                "oracle_vs_ref": int(diverging(ora, ref).sum()), "device_bit_equal_ref": int(bit_ref.sum()),
                                "device_bit_equal_oracle": int(bit_oracle.sum()), "bound": diverge_bound(name, off_ref.size)}) + "\n")
    assert int(off_oracle.sum()) <= DEVICE_VS_ORACLE, "%s: %d device samples leave the oracle's path" % (name, int(off_oracle.sum()))
    if name in PINNED:
        # the device is the oracle bit for bit, and the oracle's residual against the reference is pinned sample by sample (test_oracle_golden.PINNED)
        assert bit_oracle.all(), "%s: %d device samples are not the oracle's bit for bit" % (name, int((~bit_oracle).sum()))
        assert [[int(v) for v in c] for c in np.argwhere(~bit_ref)] == PINNED[name]
        return
    assert int(off_ref.sum()) <= diverge_bound(name, off_ref.size), "%s: %d of %d device samples differ from the reference's (measured: %d)" % (
        name, int(off_ref.sum()), off_ref.size, DIVERGING.get(name, 0))
    if name not in ULP_LEVEL:
        # bit for bit the oracle's radiance -- and therefore, outside test_oracle_golden.DIVERGING, the reference's
        assert bit_oracle.all(), "%s: %d device samples are not the oracle's bit for bit" % (name, int((~bit_oracle).sum()))
        if name not in DIVERGING:
            assert bit_ref.all(), "%s: %d device samples are not the reference's bit for bit" % (name, int((~bit_ref).sum()))
    # the few divergent paths do not move the image
    assert np.allclose(got.mean(axis=(0, 1, 2)), ref.mean(axis=(0, 1, 2)), rtol=0.03)


def test_sample_dump_covers_pass_ranges_and_shards(tmp_path):
    """samples [4, 8) of a pass equal the second half of samples [0, 8): each (pixel, sample) has its own random stream."""
    path = scenes.cornell(tmp_path, resolution=(40, 24), spp=8)
    r = tg.Renderer(path, seed=SEED)
    whole = r.trace_samples(0, 8)
    tail = r.trace_samples(4, 8)
    r.close()
    assert whole.shape == (24, 40, 8, 3) and tail.shape == (24, 40, 4, 3)
    assert (whole[:, :, 4:] == tail).all()
    assert np.isfinite(whole).all() and whole.max() > 0
