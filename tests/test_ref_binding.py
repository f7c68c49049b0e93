"""The compiled drop-in (INTEGRATION.md sections 2-4; oracle/ref_binding/, oracle/Makefile.ref `binding`): the REFERENCE's own
`tungsten` program -- Scene::load, TraceableScene, the render loop of src/tungsten/Shared.hpp, Integrator::saveOutputs -- with
"integrator": {"type": "path_tracer_hip"} registered in (a generated copy of) its IntegratorFactory table, the subclass of its
Integrator that drives libtungsten_hip.so, and the flattener that walks its TraceableScene.

* without a GPU: what that flattener makes of the reference's objects equals, array by array and byte by byte, what this
  library's own loader makes of the same JSON (the binding writes its TgHipSceneDesc to a file before it asks for a device);
* with one: the image the reference program writes is, bit for bit, the image of the stand-alone host.
"""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import scenes
import tungsten_amd as tg
from tungsten_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "oracle", "_ref", "tungsten_hip_ref")

pytestmark = pytest.mark.skipif(not os.path.exists(BINARY), reason="oracle/_ref/tungsten_hip_ref not built (make -f oracle/Makefile.ref binding)")


def hip_scene(path):
    """The same scene with the integrator's type switched to the registered plugin name."""
    d = json.load(open(path))
    d["integrator"]["type"] = "path_tracer_hip"
    out = path.replace(".json", "_hip.json")
    json.dump(d, open(out, "w"))
    return out


def run_reference(path, tmp, *args, **env):
    return subprocess.run([BINARY, "-t", "2", "-s", str(tg.DEFAULT_SEED)] + list(args) + [path], cwd=str(tmp),
                          env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)


def read_dump(path):
    out, data = {}, open(path, "rb").read()
    pos = 0
    while pos < len(data):
        n, = struct.unpack_from("<I", data, pos); pos += 4
        name = data[pos:pos + n].decode(); pos += n
        size, = struct.unpack_from("<Q", data, pos); pos += 8
        out[name] = data[pos:pos + size]; pos += size
    return out


def own_desc(path):
    flat = tg.FlattenedScene(path)
    d = flat.desc.contents

    def arr(ptr, count, size):
        return C.string_at(ptr, count*size) if count and ptr else b""
    own = {
        "nodes": arr(d.nodes, d.num_nodes, C.sizeof(capi.TgHipBvhNode)),
        "wide_nodes": arr(d.wide_nodes, d.num_wide_nodes, C.sizeof(capi.TgHipWideNode)),
        "recs": arr(d.recs, d.num_recs, C.sizeof(capi.TgHipPrimRec)),
        "tri_attrs": arr(d.tri_attrs, d.num_recs, C.sizeof(capi.TgHipTriAttr)),
        "objects": arr(d.objects, d.num_objects, C.sizeof(capi.TgHipObject)),
        "lights": arr(d.lights, d.num_lights, 4),
        "infinite_lights": arr(d.infinite_lights, d.num_infinite_lights, 4),
        "bsdfs": arr(d.bsdfs, d.num_bsdfs, C.sizeof(capi.TgHipBsdf)),
        "textures": arr(d.textures, d.num_textures, C.sizeof(capi.TgHipTexture)),
        "texels": arr(d.texels, d.num_texel_floats, 4),
        "dist": arr(d.dist, d.num_dist_floats, 4),
        "camera": bytes(d.camera),
        "settings": bytes(d.settings),
        "bounds": bytes(d.bounds_lo) + bytes(d.bounds_hi),
        "sobol": arr(d.sobol_matrices, d.num_sobol_words, 4),
        "media": arr(d.media, d.num_media, C.sizeof(capi.TgHipMedium)),
        "light_tris": arr(d.light_tris, d.num_light_tri_floats, 4),
        "inst_prims": arr(d.inst_prims, d.num_inst_prims, 4),
        "inst_leaf_boxes": arr(d.inst_leaf_boxes, d.num_inst_prims, 32),
        "inst_tight_boxes": arr(d.inst_tight_boxes, d.num_top_recs if d.num_instances else 0, 32),
        "counts": struct.pack("<II", d.num_top_recs, d.num_instances),
    }
    flat.close()
    return own


CASES = {
    "cornell": lambda tmp: scenes.cornell(tmp, resolution=(96, 54), spp=4),
    "cornell_as_shipped_sampler": lambda tmp: scenes.cornell(tmp, resolution=(96, 54), spp=4, renderer={"stratified_sampler": True, "adaptive_sampling": True}, spp_step=2),
    "materialtest": lambda tmp: scenes.materialtest(tmp, resolution=(160, 90), spp=4),
    "zoo_a": lambda tmp: scenes.GOLDEN_CASES["zoo_a"][0](tmp, **dict(scenes.GOLDEN_CASES["zoo_a"][1], resolution=(96, 54), spp=2)),
    "zoo_b": lambda tmp: scenes.GOLDEN_CASES["zoo_b"][0](tmp, **dict(scenes.GOLDEN_CASES["zoo_b"][1], resolution=(96, 54), spp=2)),
    "zoo_e": lambda tmp: scenes.GOLDEN_CASES["zoo_e"][0](tmp, **dict(scenes.GOLDEN_CASES["zoo_e"][1], resolution=(96, 54), spp=2)),
    "zoo_f": lambda tmp: scenes.GOLDEN_CASES["zoo_f"][0](tmp, **dict(scenes.GOLDEN_CASES["zoo_f"][1], resolution=(96, 54), spp=2)),
    # the procedural sky: the reference-side flattener hands over the image Tungsten's own Skydome baked, the own loader bakes it itself
    # bump-mapped bsdfs sharing one .png at several scales: the reference's TextureCache decides which scale serves them all
    "bump": lambda tmp: scenes.GOLDEN_CASES["cornell_bump"][0](tmp, **dict(scenes.GOLDEN_CASES["cornell_bump"][1], resolution=(96, 54), spp=2)),
    "skydome": lambda tmp: scenes.GOLDEN_CASES["cornell_skydome"][0](tmp, **dict(scenes.GOLDEN_CASES["cornell_skydome"][1], resolution=(96, 54), spp=2)),
}
# round 4: everything else the device renders -- the analytic primitives and emitters, mesh emitters, media with every transmittance and
# phase function on primitives and on the camera, the thin-lens camera with its three apertures, `instances` (whose master meshes the
# plugin loads itself: an unmodified `tungsten` never reads them, Instance.cpp:265-282)
for _name in ("cornell_disks", "cornell_cylinders", "cornell_point_lights", "cornell_sun_sky", "cornell_mesh_light", "cornell_mesh_and_quad_light",
              "cornell_fog", "cornell_smoke", "cornell_fog_smoke_sobol", "cornell_fog_rayleigh", "cornell_fog_davis", "cornell_fog_davis_weinstein",
              "cornell_fog_interpolated", "volumetric_caustic", "non_exponential_linear", "non_exponential_pulse", "non_exponential_erlang",
              "non_exponential_double_exponential", "non_exponential_quadratic", "cornell_thinlens", "cornell_thinlens_cateye", "cornell_thinlens_blade5",
              "cornell_thinlens_pivot", "cornell_thinlens_bitmap", "cornell_instances", "cornell_expfog", "cornell_expfog_smoke_sobol", "cornell_atmosphere", "cornell_atmosphere_smoke_sobol", "cornell_equirectangular", "cornell_cubemap_cross"):
    CASES[_name] = (lambda n: lambda tmp: scenes.GOLDEN_CASES[n][0](tmp, **dict(scenes.GOLDEN_CASES[n][1], resolution=(96, 54), spp=2)))(_name)
WIDENED = ["cornell_disks", "cornell_cylinders", "cornell_point_lights", "cornell_sun_sky", "cornell_mesh_light", "cornell_fog_smoke_sobol", "volumetric_caustic",
           "non_exponential_pulse", "cornell_thinlens_blade5", "cornell_thinlens_cateye", "cornell_instances", "zoo_e", "cornell_expfog_smoke_sobol", "cornell_atmosphere_smoke_sobol", "cornell_equirectangular", "cornell_cubemap_cross"]


@pytest.mark.parametrize("case", sorted(CASES))
def test_reference_side_flattener_builds_the_scene_the_own_loader_builds(case, tmp_path):
    if case == "materialtest" and not scenes.have_materialtest():
        pytest.skip("materialtest assets not present")
    path = CASES[case](str(tmp_path))
    dump = os.path.join(str(tmp_path), "desc.bin")
    r = run_reference(hip_scene(path), tmp_path, TGHIP_REF_DUMP_DESC=dump)
    assert os.path.exists(dump), r.stdout
    ref, own = read_dump(dump), own_desc(path)
    assert sorted(ref) == sorted(own)
    for name in sorted(own):
        assert len(ref[name]) == len(own[name]), "%s: %d bytes from the reference-side flattener, %d from the own loader" % (name, len(ref[name]), len(own[name]))
        if ref[name] != own[name]:
            a, b = np.frombuffer(ref[name], np.uint8), np.frombuffer(own[name], np.uint8)
            first = int(np.nonzero(a != b)[0][0])
            raise AssertionError("%s differs in %d of %d bytes, first at byte %d" % (name, int((a != b).sum()), a.size, first))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cornell", "cornell_as_shipped_sampler", "materialtest", "skydome", "bump"] + WIDENED)
def test_reference_program_with_the_plugin_writes_the_image_of_the_own_host(case, tmp_path):
    """tungsten (the reference's program) with "type": "path_tracer_hip" against tungsten_hip (this repository's CLI) on the same
    scene, seed and spp: the .pfm files are identical bit for bit."""
    if case == "materialtest" and not scenes.have_materialtest():
        pytest.skip("materialtest assets not present")
    path = CASES[case](str(tmp_path))
    ref_pfm, own_pfm = os.path.join(str(tmp_path), "ref.pfm"), os.path.join(str(tmp_path), "own.pfm")
    r = run_reference(hip_scene(path), tmp_path, "-e", ref_pfm, "-o", os.path.join(str(tmp_path), "ref.png"))
    assert r.returncode == 0 and os.path.exists(ref_pfm), r.stdout
    cli = os.path.join(ROOT, "tungsten_amd", "lib", "tungsten_hip")
    o = subprocess.run([cli, "-s", str(tg.DEFAULT_SEED), "-e", own_pfm, "-o", os.path.join(str(tmp_path), "own.png"), path], cwd=str(tmp_path),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert o.returncode == 0 and os.path.exists(own_pfm), o.stdout
    a, b = open(ref_pfm, "rb").read(), open(own_pfm, "rb").read()
    assert len(a) == len(b)
    assert a == b, "the two programs' images differ in %d bytes" % sum(x != y for x, y in zip(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("adaptive", [False, True])
def test_reference_program_resumes_a_render_through_the_plugin(adaptive, tmp_path):
    """`tungsten` with "enable_resume_render" (src/tungsten/Shared.hpp:256-320; Integrator::saveRenderResumeData / resumeRender,
    integrators/Integrator.cpp:108-162) and the plugin's saveState / loadState: half the samples, exit, a second run that resumes from the
    state file and renders the rest -- the .pfm of the uninterrupted render bit for bit (the state holds the device's radiance sums, the
    SampleRecords and the scheduler's sampler; with adaptive sampling the second run's sample distribution depends on all three)."""
    import json
    import shutil
    import tempfile
    # (a directory with a SHORT path: the reference's own resume -- with its own path_tracer too -- answers "Resume unsuccessful" whenever the
    # scene's directory path is longer than 15 characters, pytest's tmp_path always is; found while writing this test, not looked into)
    tmp_path = tempfile.mkdtemp(prefix="r", dir="/tmp")
    try:
        _resume_case(adaptive, tmp_path)
    finally:
        shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.parametrize("adaptive", [False, True])
def test_resume_plumbing_of_the_plugin_without_a_device(adaptive):
    """The same two runs with TGHIP_REF_DRY_RUN (the plugin creates no context and a pass renders nothing): Integrator::saveRenderResumeData ->
    saveState, the state file, resumeRender -> loadState and the pass loop that continues at the saved spp, on a box without a GPU."""
    import shutil
    import tempfile
    tmp_path = tempfile.mkdtemp(prefix="r", dir="/tmp")
    try:
        _resume_case(adaptive, tmp_path, TGHIP_REF_DRY_RUN="1")
    finally:
        shutil.rmtree(tmp_path, ignore_errors=True)


def _resume_case(adaptive, tmp_path, **env):
    import json
    kw = dict(resolution=(96, 54), spp=32, spp_step=16)
    if adaptive:
        kw["renderer"] = {"adaptive_sampling": True, "stratified_sampler": True}
    base = hip_scene(scenes.cornell(str(tmp_path), **kw))
    whole = os.path.join(str(tmp_path), "whole.pfm")
    r = run_reference(base, tmp_path, "-e", whole, "-o", os.path.join(str(tmp_path), "whole.png"), **env)
    assert r.returncode == 0 and os.path.exists(whole), r.stdout
    d = json.load(open(base))
    d["renderer"].update(enable_resume_render=True, resume_render_file="state.dat")
    resumable = os.path.join(str(tmp_path), "resumable.json")
    json.dump(d, open(resumable, "w"))
    part = os.path.join(str(tmp_path), "part.pfm")
    r1 = run_reference(resumable, tmp_path, "--spp", "16", "-e", part, "-o", os.path.join(str(tmp_path), "part.png"), **env)
    assert r1.returncode == 0 and os.path.exists(os.path.join(str(tmp_path), "state.dat")), r1.stdout
    r2 = run_reference(resumable, tmp_path, "-e", part, "-o", os.path.join(str(tmp_path), "part.png"), **env)
    assert r2.returncode == 0 and "Resume successful" in r2.stdout, r2.stdout
    assert "Completed 32/32 spp" in r2.stdout and "Completed 16/32 spp" not in r2.stdout, r2.stdout
    a, b = open(whole, "rb").read(), open(part, "rb").read()
    assert a == b, "the resumed render differs from the uninterrupted one in %d bytes" % sum(x != y for x, y in zip(a, b))
