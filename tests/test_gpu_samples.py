"""Per-sample parity of the HIP path: the radiance of every individual (pixel, sample) -- what PathTracer::traceSample
returns (integrators/path_tracer/PathTracer.cpp:14-149) -- read back through TGHIP_PASS_SAMPLES / tghip_download_samples and
compared with the reference's own per-sample output (tests/golden/*_samples.npz, rendered by oracle/ref_harness.cpp with the
shared counter-based random stream), by the metric tests/test_oracle_golden.py holds the oracle to: a sample agrees when every
channel is within 1e-3 of the reference (relative to the sample's largest channel); a path is a chaotic function of its hits, so
an ulp-level difference at an edge or a coin flip sends it elsewhere, and the fraction of samples allowed to do so is stated
per case (DEVICE_DIVERGE; the measured fractions are tabulated in DESIGN.md section 7)."""
import json
import os

import numpy as np
import pytest

import oracle_lib
import scenes
import tungsten_amd as tg
from test_oracle_golden import DIVERGE as ORACLE_DIVERGE

pytestmark = pytest.mark.gpu
SEED = tg.DEFAULT_SEED
TABLE = os.environ.get("TG_DIVERGE_TABLE")     # when set: append one JSON line per case (tools/gpu_session scripts)

# Fraction of device samples allowed to leave the reference's path.  The device evaluates glibc's own sinf / cosf / logf / expf / acosf
# (csrc/hip/pt_libm.h, pt_math.h: the algorithms restated and matched bit for bit, tests/test_gpu_libm.py) and differs from the oracle's
# arithmetic in atan2f / powf / cbrtf (ocml), which is where paths fork beyond the oracle's own forks (coincident surfaces, Embree's
# rcp + Newton division): each bound below is the oracle's bound for the case (tests/test_oracle_golden.py: DIVERGE) plus a margin for those.
def device_bound(name):
    if name == "non_exponential_davis":
        # The 4.7 x 3.8 mm emitters of the shipped scene make chooseLight's weights (Quad::approximateRadiance, Quad.cpp:253-281: 2 pi minus
        # four arc cosines) a magnifier for the last bit of everything upstream: with ocml's acosf a third of the samples of the six
        # non-exponential cases landed outside 1e-3 (round 2), with glibc's acosf restated 0.9-2.1 %, with sinf / cosf / logf / expf
        # restated as well NONE in five of the six cases -- and 0.39 % in this one, whose Davis transmittance calls powf (still ocml's).
        return 0.01
    return max(3.0*ORACLE_DIVERGE.get(name, 0.0), 2e-3)


def _skip(name):
    if name == "water_caustic" and not scenes.have_water_caustic():
        pytest.skip("water-caustic assets (oracle/_ref/data) not present")
    if ("materialtest" in name or name == "mesh1m") and not scenes.have_materialtest():
        pytest.skip("materialtest assets (oracle/_ref/data) not present")


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
    bad = diverging(got, ref)
    frac = float(bad.mean())
    if TABLE:
        with open(TABLE, "a") as f:
            f.write(json.dumps({"case": name, "samples": int(bad.size), "device_diverging": int(bad.sum()), "device_frac": frac,
                                "oracle_bound": ORACLE_DIVERGE.get(name, 0.0), "device_bound": device_bound(name)}) + "\n")
    assert frac <= device_bound(name), "%s: %.4f%% of the device's samples differ from the reference's" % (name, 100*frac)
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
