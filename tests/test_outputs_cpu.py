"""Auxiliary output buffers (cameras/OutputBuffer.hpp, PathTracer.cpp:78-96, 133-140), CPU side: the oracle's
restatement against the buffers the reference's OWN integrator loop filled (tests/golden/zoo_a_outputs_integrate.npz, written
by `ref_harness integrate` = Camera::serializeOutputBuffers)."""
import os

import numpy as np

import oracle_lib
import scenes
import tungsten_amd as tg

NAME = "zoo_a_outputs"
CH = {"color": slice(0, 3), "depth": slice(3, 4), "normal": slice(4, 7), "albedo": slice(7, 10), "visibility": slice(10, 11)}


def gold():
    return np.load(os.path.join(scenes.GOLDEN, NAME + "_integrate.npz"))


def combined_mean(a, b, count_per_channel):
    """OutputBuffer::operator[] (OutputBuffer.hpp:134-144)."""
    n = count_per_channel.astype(np.float64)
    na, nb = np.floor((n + 1)/2), np.floor(n/2)
    return (a*na + b*nb)/np.maximum(n, 1)


def per_channel(count):
    """[H, W, 5] output counts -> [H, W, 11] channel counts."""
    return np.concatenate([np.repeat(count[..., i:i + 1], c.stop - c.start, axis=2) for i, c in enumerate(CH.values())], axis=2)


def check_exactly_against_gold(aux_a, aux_b, aux_var, aux_count, g):
    """Since the end of round 4 (every sample on the reference's path): the counts, the A / B halves and the variance sums of
    all five outputs are the reference's in every pixel, float32 bit for bit."""
    assert (aux_count == g["aux_count"]).all()
    for name, mine, theirs in (("a", aux_a, g["aux_a"]), ("b", aux_b, g["aux_b"]), ("variance", aux_var, g["aux_variance"])):
        same = np.asarray(mine, np.float32).view(np.uint32) == np.asarray(theirs, np.float32).view(np.uint32)
        for out, sl in CH.items():
            assert same[..., sl].all(), (name, out, float(same[..., sl].all(axis=-1).mean()))


def check_against_gold(aux_a, aux_b, aux_var, aux_count, g, frac_ok=0.97):
    """Counts of the outputs every sample records (or deterministically not) exactly up to divergent paths; values per pixel."""
    gc = g["aux_count"]
    assert (aux_count[..., 0] == gc[..., 0]).all()                              # colour: every sample
    for i, name in enumerate(CH):
        same = (aux_count[..., i] == gc[..., i]).mean()
        assert same >= frac_ok, (name, same)
    cc, gcc = per_channel(aux_count), per_channel(gc)
    mean, gmean = combined_mean(aux_a, aux_b, cc), combined_mean(g["aux_a"], g["aux_b"], gcc)
    for name, sl in CH.items():
        scale = np.abs(gmean[..., sl]).max() + 1e-6
        close = np.isclose(mean[..., sl], gmean[..., sl], rtol=5e-3, atol=2e-3*scale).all(axis=-1)
        assert close.mean() >= frac_ok, (name, close.mean())
        assert np.allclose(mean[..., sl].mean(axis=(0, 1)), gmean[..., sl].mean(axis=(0, 1)), rtol=2e-2, atol=2e-3*scale), name
        # the A / B halves and the variance sums individually
        for mine, theirs in ((aux_a, g["aux_a"]), (aux_b, g["aux_b"]), (aux_var, g["aux_variance"])):
            s = np.abs(theirs[..., sl]).max() + 1e-6
            ok = np.isclose(mine[..., sl], theirs[..., sl], rtol=2e-2, atol=5e-3*s).all(axis=-1)
            assert ok.mean() >= frac_ok - 0.02, (name, ok.mean())


def test_oracle_output_buffers_match_the_reference(tmp_path):
    g = gold()
    mk, kw = scenes.OUTPUT_CASES[NAME]
    flat = tg.FlattenedScene(mk(tmp_path, name=NAME + ".json", **kw))
    w, h = flat.width, flat.height
    ssum, count, rec, pass_spp, aux = oracle_lib.integrate_aux(flat.desc, w, h, int(g["seed"]), kw["spp"], kw["spp_step"], False, True)
    flat.close()
    assert len(rec) == len(g["records"])
    check_against_gold(aux["a"], aux["b"], aux["variance"], aux["count"], g)
    check_exactly_against_gold(aux["a"], aux["b"], aux["variance"], aux["count"], g)
    # the colour output is the framebuffer: same sample counts, same mean
    assert (aux["count"][..., 0] == count).all()
    cm = combined_mean(aux["a"][..., :3], aux["b"][..., :3], np.repeat(count[..., None], 3, axis=2))
    assert np.allclose(cm, ssum/np.maximum(count, 1)[..., None], rtol=1e-4, atol=1e-6)
