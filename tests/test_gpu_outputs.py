"""Auxiliary output buffers on the HIP path (TGHIP_PASS_AUX): against the buffers of the reference's own integrator loop
(tests/golden/zoo_a_outputs_integrate.npz), against the oracle, and structural properties at full size."""
import os

import numpy as np
import pytest

import oracle_lib
import scenes
import tungsten_amd as tg
from test_outputs_cpu import CH, NAME, check_against_gold, check_exactly_against_gold, combined_mean, gold, per_channel

pytestmark = pytest.mark.gpu


def _render(path, seed):
    r = tg.Renderer(path, seed=seed)
    r.render()
    mean, ssum, count = r.image()
    aux = r.output_buffers()
    r.close()
    return mean, ssum, count, aux


def test_gpu_output_buffers_match_reference_and_oracle(tmp_path):
    g = gold()
    mk, kw = scenes.OUTPUT_CASES[NAME]
    path = mk(tmp_path, name=NAME + ".json", **kw)
    mean, ssum, count, aux = _render(path, int(g["seed"]))
    check_against_gold(aux["a"], aux["b"], aux["variance"], aux["count"], g, frac_ok=0.96)
    check_exactly_against_gold(aux["a"], aux["b"], aux["variance"], aux["count"], g)
    flat = tg.FlattenedScene(path)
    osum, ocount, rec, pass_spp, oaux = oracle_lib.integrate_aux(flat.desc, flat.width, flat.height, int(g["seed"]), kw["spp"], kw["spp_step"], False, True)
    flat.close()
    for i, name in enumerate(CH):
        assert (aux["count"][..., i] == oaux["count"][..., i]).mean() >= 0.97, name
    m, om = combined_mean(aux["a"], aux["b"], per_channel(aux["count"])), combined_mean(oaux["a"], oaux["b"], per_channel(oaux["count"]))
    for name, sl in CH.items():
        scale = np.abs(om[..., sl]).max() + 1e-6
        assert np.isclose(m[..., sl], om[..., sl], rtol=5e-3, atol=2e-3*scale).all(axis=-1).mean() >= 0.96, name
    # the colour output is the framebuffer
    assert (aux["count"][..., 0] == count).all()
    assert np.allclose(combined_mean(aux["a"][..., :3], aux["b"][..., :3], np.repeat(count[..., None], 3, axis=2)), mean, rtol=2e-4, atol=1e-6)


def test_gpu_output_buffers_full_size_properties(tmp_path):
    """1280x720, 16 spp in two passes, uniform sampler: what must hold whatever the scene."""
    spp = 16
    path = scenes.cornell(tmp_path, name="outputs_full.json", resolution=(1280, 720), spp=spp, spp_step=8, edit=scenes._outputs)
    mean, ssum, count, aux = _render(path, tg.DEFAULT_SEED)
    c = aux["count"]
    assert (count == spp).all() and (c[..., 0] == spp).all()
    assert (c[..., 1:] <= spp).all()
    assert (c[..., 2] >= c[..., 1]).all()                  # a sample that records a depth records a normal
    assert (c[..., 4] <= c[..., 3]).all()                  # visibility only where a surface vertex recorded
    m = combined_mean(aux["a"], aux["b"], per_channel(c))
    assert np.isfinite(m).all() and np.isfinite(aux["variance"]).all() and (aux["variance"] >= -1e-6).all()
    assert (m[..., 3] >= 0).all()                          # depth
    assert (np.linalg.norm(m[..., 4:7], axis=-1) <= 1 + 1e-4).all()   # a mean of unit normals
    assert (m[..., 10] >= 0).all() and (m[..., 10] <= 1 + 1e-6).all()  # visibility = mean transmittance
    assert np.allclose(m[..., :3], mean, rtol=2e-4, atol=1e-6)
    # the two halves are independent estimates of the same image
    a, b = aux["a"][..., :3].mean(axis=(0, 1)), aux["b"][..., :3].mean(axis=(0, 1))
    assert np.allclose(a, b, rtol=2e-2)
    # the floor of the Cornell box is seen directly through most of the lower image: constant albedo there
    assert np.allclose(m[700, 300:900, 7:10], m[700, 600, 7:10][None], rtol=1e-3)


def test_gpu_output_files_are_written(tmp_path):
    def edit(scene):
        scenes._outputs(scene)
        for b in scene["renderer"]["output_buffers"]:
            b["hdr_output_file"] = b["type"] + ".pfm"
            b["ldr_output_file"] = b["type"] + ".png"
    path = scenes.cornell(tmp_path, name="outfiles.json", resolution=(64, 36), spp=8, edit=edit)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        r = tg.Renderer(path, seed=tg.DEFAULT_SEED)
        r.render()
        r.save_outputs()
        aux = r.output_buffers()
        r.close()
    finally:
        os.chdir(cwd)
    for t in scenes.OUTPUT_TYPES:                          # OutputBuffer::save (OutputBuffer.hpp:146-189): file, fileA, fileB, fileVariance
        for tag in ("", "A", "B", "Variance"):
            for ext in (".pfm", ".png"):
                assert os.path.exists(os.path.join(str(tmp_path), t + tag + ext)), t + tag + ext
    depth_a = tg.load_pfm(os.path.join(str(tmp_path), "depthA.pfm"))
    assert np.allclose(depth_a.reshape(36, 64, -1)[..., 0], aux["a"][..., 3], rtol=1e-6)


def test_resume_keeps_the_output_buffers(tmp_path):
    """Camera::serializeOutputBuffers / deserializeOutputBuffers inside the resume file (Integrator.cpp:118-121, 152-155):
    render two of four passes, save, resume in a new renderer, finish -- every output buffer equals the uninterrupted
    render bit for bit (each pixel's running means continue where they stopped)."""
    state = str(tmp_path/"state.dat")
    rend = {"adaptive_sampling": True, "stratified_sampler": True, "enable_resume_render": True, "resume_render_file": state}
    kw = dict(resolution=(70, 42), spp=64, spp_step=16, renderer=rend, edit=scenes._outputs)
    path = scenes.cornell(tmp_path, **kw)
    full = _render(path, tg.DEFAULT_SEED)

    r = tg.Renderer(path, seed=tg.DEFAULT_SEED)
    r.step(); r.step()
    r.save_resume_data()
    r.close()
    r = tg.Renderer(path, seed=tg.DEFAULT_SEED)
    assert r.resume() and r.current_spp == 32
    while not r.step():
        pass
    mean, ssum, count = r.image()
    aux = r.output_buffers()
    r.close()
    assert ssum.tobytes() == full[1].tobytes() and (count == full[2]).all()
    assert aux.tobytes() == full[3].tobytes()
