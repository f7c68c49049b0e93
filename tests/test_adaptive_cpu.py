"""Sobol' sampler + adaptive sampling, CPU side: the host's pass scheduler
(tungsten_amd/csrc/host/Sampling.cpp) and the oracle's restatement against dumps of the reference's OWN
PathTraceIntegrator loop (tests/golden/*_integrate.npz, written by `ref_harness integrate` through tools/make_golden.py).

Integer outputs (tile seeds, sample indices, per-record sample counts) are compared exactly; adaptive weights are
floats computed by the same IEEE operations in the same order and are compared exactly as well."""
import os

import numpy as np
import pytest

import oracle_lib
import scenes
import tungsten_amd as tg

G = scenes.GOLDEN


def _gold(name):
    if "materialtest" in name and not scenes.have_materialtest():
        pytest.skip("materialtest assets (assets/) not present")
    return np.load(os.path.join(G, name + "_integrate.npz"))


def _case(name, tmp_path):
    mk, kw = scenes.INTEGRATE_CASES[name]
    return mk(tmp_path, name=name + ".json", **kw), kw


def test_sobol_matrices_table():
    m = tg.sobol_matrices()
    assert m.shape == (1024, 52) and m.dtype == np.uint32
    # dimension 0 is the van der Corput sequence; every matrix is upper triangular with a unit diagonal in bit order
    assert [int(v) for v in m[0, :32]] == [1 << (31 - i) for i in range(32)]
    for d in (1, 2, 3, 17, 500, 1023):
        for i in range(32):
            assert (int(m[d, i]) >> (31 - i)) & 1 == 1 and int(m[d, i]) & ((1 << (31 - i)) - 1) == 0, (d, i)


def test_sobol_first_points_are_stratified():
    """Unscrambled, the first 2^k points of every dimension hit every interval of width 2^-k exactly once."""
    m = tg.sobol_matrices()
    for d in (0, 1, 2, 5, 40, 1023):
        vals = []
        for index in range(64):
            r, i, col = 0, index, 0
            while i:
                if i & 1:
                    r ^= int(m[d, col])
                i >>= 1
                col += 1
            vals.append(r >> 26)
        assert sorted(vals) == list(range(64)), d


@pytest.mark.parametrize("name", sorted(scenes.INTEGRATE_CASES))
def test_tile_seeds_match_reference(name):
    g = _gold(name)
    h, w = g["image"].shape[:2]
    sch = tg.PassScheduler(w, h, int(g["seed"]))
    seeds = sch.tile_seeds
    assert sch.num_tiles == ((w + 15)//16)*((h + 15)//16) and sch.num_records == ((w + 3)//4)*((h + 3)//4)
    assert (oracle_lib.dice_tiles(w, h, int(g["seed"]))[0] == seeds).all()
    if int(g["sobol"]):     # the dump holds the SobolPathSampler seeds the reference's diceTiles drew
        assert (seeds == g["tile_seeds"]).all()


@pytest.mark.parametrize("which", ["host", "oracle"])
@pytest.mark.parametrize("name", sorted(scenes.INTEGRATE_CASES))
def test_generate_work_matches_reference(name, which, tmp_path):
    """Feed the reference's per-pass record statistics into the scheduler: sample indices, per-record sample counts and
    adaptive weights of the next pass must come out exactly as the reference's generateWork produced them."""
    g = _gold(name)
    _, kw = _case(name, tmp_path)
    h, w = g["image"].shape[:2]
    spp, step = kw["spp"], kw["spp_step"]
    seed = int(g["seed"])
    if which == "host":
        sch = tg.PassScheduler(w, h, seed)
        rec = sch.records
    else:
        rec = np.zeros(((w + 3)//4)*((h + 3)//4), oracle_lib.RECORD_DTYPE)
        state = oracle_lib.dice_tiles(w, h, seed)[1]
    cur = 0
    saw_adaptive = False
    for p, gold in enumerate(g["records"]):
        nxt = min(cur + step, spp)
        gold = gold.ravel()
        if which == "host":
            has_work = sch.generate_work(cur, nxt, True)
        else:
            has_work, state = oracle_lib.generate_work(rec, w, h, state, cur, nxt, True)
        assert has_work
        assert int(g["pass_spp"][p]) == nxt
        for f in ("sample_index", "next_sample_count", "adaptive_weight"):
            assert (rec[f] == gold[f]).all(), (p, f)
        saw_adaptive |= bool((gold["next_sample_count"] != nxt - cur).any())
        # what the device (or the oracle's renderer) would hand back after rendering the pass
        for f in ("sample_count", "mean", "running_variance"):
            rec[f] = gold[f]
        cur = nxt
    assert saw_adaptive, "the case never reached the adaptive branch"
    # budget of an adaptive pass (PathTraceIntegrator.cpp:93-95): about (spp_step - 1)*W*H/16 extra samples per record set
    last = g["records"][-1]["next_sample_count"].astype(np.int64)
    assert abs(int(last.sum()) - ((spp - int(g["pass_spp"][-2]) - 1)*w*h//16 + last.size)) <= last.size


def test_generate_work_without_adaptive_is_uniform():
    sch = tg.PassScheduler(40, 24, 7)
    assert sch.generate_work(0, 16, False)
    assert (sch.records["next_sample_count"] == 16).all() and (sch.records["sample_index"] == 0).all()
    assert sch.generate_work(16, 24, False)
    assert (sch.records["next_sample_count"] == 8).all() and (sch.records["sample_index"] == 16).all()


def test_generate_work_reports_no_work_when_converged():
    """errorPercentile95() == 0 (every record has zero variance) => generateWork returns false (PathTraceIntegrator.cpp:118-121)."""
    sch = tg.PassScheduler(16, 16, 1)
    assert sch.generate_work(0, 16, True)
    rec = sch.records
    rec["sample_count"] = 256
    rec["mean"] = 0.5
    rec["running_variance"] = 0.0
    assert not sch.generate_work(16, 32, True)
    assert (rec["sample_index"] == 16).all()


# (Until the end of round 4 later passes were allowed to drift -- cornell_adaptive after three, the Sobol' case after two passes, materialtest's
# statistics within 1e-3: a single path on the other side of a coincident-face tie perturbs the stochastic rounding of every following record.
# With the reference's top-level tree restated, DESIGN.md section 8, no path is: every pass of every case is exact, statistics bit for bit.)
@pytest.mark.parametrize("name,exact_passes,min_equal,stats_rtol", [("cornell_adaptive", 5, 1.0, 0.0), ("cornell_adaptive_sobol", 4, 1.0, 0.0),
                                                                     ("materialtest_as_shipped", 3, 1.0, 0.0)])
def test_oracle_integrate_matches_reference(name, exact_passes, min_equal, stats_rtol, tmp_path):
    """The oracle's whole pass loop (tile seeds -> render with Welford records -> generateWork -> ...) against the
    reference's: every pass reproduces the reference's sample schedule, sample counts and Welford statistics exactly."""
    g = _gold(name)
    path, kw = _case(name, tmp_path)
    flat = tg.FlattenedScene(path)
    w, h = flat.width, flat.height
    assert bool(flat.info.adaptive_sampling) and bool(flat.info.stratified_sampler) == bool(int(g["sobol"]))
    ssum, count, rec, pass_spp = oracle_lib.integrate(flat.desc, w, h, int(g["seed"]), kw["spp"], kw["spp_step"], True, bool(int(g["sobol"])))
    flat.close()
    assert (pass_spp == g["pass_spp"]).all()
    gold = g["records"]
    assert len(gold) == exact_passes
    for p in range(len(gold)):
        same_schedule = (rec[p]["next_sample_count"] == gold[p]["next_sample_count"]).mean()
        # Welford mean / running variance: bit-equal on the Cornell box (every sample is), within stats_rtol on materialtest
        # (triangle hits differ from Embree's in the last bits, tests/test_oracle_golden.py)
        same_stats = (np.isclose(rec[p]["mean"], gold[p]["mean"], rtol=stats_rtol, atol=0.0) &
                      np.isclose(rec[p]["running_variance"], gold[p]["running_variance"], rtol=10*stats_rtol, atol=0.0)).mean()
        if p < exact_passes:
            assert same_schedule == 1.0 and (rec[p]["sample_index"] == gold[p]["sample_index"]).all(), p
            assert (rec[p]["sample_count"] == gold[p]["sample_count"]).all(), p
        if p < exact_passes:
            assert same_stats == 1.0, (p, same_stats)
        assert same_schedule >= min_equal, (p, same_schedule)
        # the total number of samples of a pass is fixed by the budget, whatever the distribution
        assert abs(int(rec[p]["next_sample_count"].astype(np.int64).sum()) - int(gold[p]["next_sample_count"].astype(np.int64).sum())) <= 2
    mean = ssum/np.maximum(count, 1)[..., None]
    assert np.allclose(mean.mean(axis=(0, 1)), g["image"].mean(axis=(0, 1)), rtol=0.02)
    lum = lambda a: a @ np.array([0.2126, 0.7152, 0.0722])
    # per-pixel agreement where the schedule is identical is tested above through the records; globally: small RMS
    assert np.sqrt(((lum(mean) - lum(g["image"]))**2).mean()) <= 0.15*lum(g["image"]).mean() + 1e-3
