"""Generates tests/golden/ by running the REFERENCE ITSELF (oracle/_ref/, built from /root/reference by
oracle/Makefile.ref).  TEST INFRASTRUCTURE; run in the build container only (needs oracle/_ref):

    python tools/make_golden.py

Outputs (all small, committed):
  <case>_samples.npz   radiance of every individual sample, float32[h][w][spp][3], from the reference's own
                       PathTracer::traceSample driven by oracle/ref_harness.cpp with the counter-based
                       random stream (seed, pixel, sample) that oracle.c and the HIP kernels also use; `*_sobol` cases
                       ("stratified_sampler": true) draw next1D/next2D from the tiles' SobolPathSampler sequence
  <case>_integrate.npz the SampleRecords after every pass (sampleCount, nextSampleCount, sampleIndex, adaptiveWeight,
                       mean, runningVariance) and the final image of the reference's OWN PathTraceIntegrator loop
                       (diceTiles, generateWork, adaptive sampling, OutputBuffer), tile samplers swapped for the
                       counter-based ones; plus the tile seeds diceTiles drew
  <scene>_units.json   known answers of the deterministic building blocks (rng, camera rays, closest hits,
                       bsdf eval/pdf/sample, light sampleDirect/directPdf/evalDirect)
  <scene>_converged.npz  mean image of the unmodified reference binary (`tungsten -s <seed>`, its own
                       per-tile sampler) at high spp: the statistical anchor
The scene variants are produced by tests/scenes.py, which the tests call again with the same arguments.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import tungsten_amd as tg  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
HARNESS = os.path.join(REF, "ref_harness")
TUNGSTEN = os.path.join(REF, "tungsten")
SEED = tg.DEFAULT_SEED


def samples(path, w, h, spp, out):
    tmp = out + ".bin"
    subprocess.check_call([HARNESS, "samples", path, str(SEED), str(spp), tmp], stdout=subprocess.DEVNULL)
    a = np.fromfile(tmp, np.float32).reshape(h, w, spp, 3)
    os.remove(tmp)
    np.savez_compressed(out, samples=a, seed=np.uint32(SEED))
    print("%-40s %s mean %s" % (os.path.basename(out), a.shape, a.mean(axis=(0, 1, 2))))


def integrate(path, out):
    """SampleRecords after every pass + final image of the reference's own PathTraceIntegrator loop."""
    tmp = out + ".bin"
    subprocess.check_call([HARNESS, "integrate", path, str(SEED), tmp], cwd=os.path.dirname(path), stdout=subprocess.DEVNULL)
    b = open(tmp, "rb").read()
    os.remove(tmp)
    w, h, vw, vh, passes, sobol = np.frombuffer(b, np.uint32, 6, 0)
    tiles = ((w + 15)//16)*((h + 15)//16)
    off = 24
    tile_seeds = np.frombuffer(b, np.uint32, tiles, off).copy()
    off += tiles*4
    rec_dtype = np.dtype([("sample_count", "<u4"), ("next_sample_count", "<u4"), ("sample_index", "<u4"),
                          ("adaptive_weight", "<f4"), ("mean", "<f4"), ("running_variance", "<f4")])
    pass_spp, records = [], []
    for _ in range(passes):
        pass_spp.append(np.frombuffer(b, np.uint32, 1, off)[0])
        off += 4
        records.append(np.frombuffer(b, rec_dtype, vw*vh, off).reshape(vh, vw).copy())
        off += vw*vh*rec_dtype.itemsize
    image = np.frombuffer(b, np.float32, w*h*3, off).reshape(h, w, 3).copy()
    off += image.nbytes
    extra = {}
    if off < len(b):
        # renderer.output_buffers (all five outputs, two_buffer_variance and sample_variance on): Camera::serializeOutputBuffers,
        # per output _bufferA, _bufferB, _variance, _sampleCount -> channel layout of TgHipAuxPixel (include/tungsten_hip.h)
        n = int(w)*int(h)
        a, bb, var, cnt = [], [], [], []
        for ch in (3, 1, 3, 3, 1):                      # color, depth, normal, albedo, visibility
            for dst in (a, bb, var):
                dst.append(np.frombuffer(b, np.float32, n*ch, off).reshape(h, w, ch).copy())
                off += n*ch*4
            cnt.append(np.frombuffer(b, np.uint32, n, off).reshape(h, w, 1).copy())
            off += n*4
        extra = dict(aux_a=np.concatenate(a, axis=2), aux_b=np.concatenate(bb, axis=2), aux_variance=np.concatenate(var, axis=2),
                     aux_count=np.concatenate(cnt, axis=2))
    assert off == len(b)
    np.savez_compressed(out, tile_seeds=tile_seeds, pass_spp=np.array(pass_spp, np.uint32), records=np.stack(records), image=image,
                        sobol=np.uint32(sobol), seed=np.uint32(SEED), **extra)
    print("%-40s %d passes, counts of the last pass %d..%d, image mean %s" % (
        os.path.basename(out), passes, records[-1]["next_sample_count"].min(), records[-1]["next_sample_count"].max(), image.mean(axis=(0, 1))))


def units(path, out):
    subprocess.check_call([HARNESS, "units", path, out], cwd=os.path.dirname(path), stdout=subprocess.DEVNULL)
    print("%-40s %d bytes" % (os.path.basename(out), os.path.getsize(out)))


def converged(path, spp, out, tmp):
    stem = os.path.join(tmp, "conv_" + os.path.basename(out))   # unique: the reference renames instead of overwriting
    pfm = stem + ".pfm"
    subprocess.check_call([TUNGSTEN, "-t", str(os.cpu_count()), "-s", str(SEED), "--spp", str(spp), "-e", pfm, "-o",
                           stem + ".png", path], stdout=subprocess.DEVNULL, cwd=tmp)
    img = tg.load_pfm(pfm)
    np.savez_compressed(out, mean=img, spp=np.uint32(spp))
    print("%-40s %s mean %s" % (os.path.basename(out), img.shape, img.mean(axis=(0, 1))))


def as_shipped_converged(tmp, g):
    """materialtest AS IT SHIPS -- the Sobol' sampler and adaptive sampling in 16-spp passes -- rendered by the unmodified reference binary at 256x144, 256 spp:
    the L2 anchor of the Sobol' / supplemental-stream draw order and of the pass loop (every other comparison of that configuration is against the
    reference with this library's sample stream injected)."""
    converged(scenes.materialtest(tmp, name="c_materialtest_as_shipped.json", resolution=(256, 144), spp=256, spp_step=16,
                                  renderer={"adaptive_sampling": True, "stratified_sampler": True}), 256, g("materialtest_as_shipped_converged.npz"), tmp)


def main():
    if not os.path.exists(HARNESS):
        raise SystemExit("oracle/_ref/ref_harness missing: run `python -c 'import __graft_entry__ as g; g.build()'` where /root/reference exists")
    os.makedirs(scenes.GOLDEN, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="tg_golden_")
    g = lambda n: os.path.join(scenes.GOLDEN, n)
    try:
        if sys.argv[1:] == ["as_shipped_converged"]:
            as_shipped_converged(tmp, g)
            return
        if len(sys.argv) > 1:
            # python tools/make_golden.py case ...: the per-sample goldens of the named cases only (and the unit answers of named zoo scenes)
            for name in sys.argv[1:]:
                mk, kw = scenes.GOLDEN_CASES[name] if name in scenes.GOLDEN_CASES else scenes.LIFTED_CASES[name]
                p = mk(tmp, name=name + ".json", **kw)
                with open(p) as f:
                    sc = json.load(f)
                w, h = sc["camera"]["resolution"]
                samples(p, w, h, sc["renderer"]["spp"], g(name + "_samples.npz"))
                if name in scenes.ZOO:
                    units(scenes.cornell_zoo(tmp, name, name="u_%s.json" % name, resolution=(96, 54), spp=1), g(name + "_units.json"))
            return
        for name, (mk, kw) in list(scenes.GOLDEN_CASES.items()) + list(scenes.LIFTED_CASES.items()):
            p = mk(tmp, name=name + ".json", **kw)
            with open(p) as f:
                sc = json.load(f)
            w, h = sc["camera"]["resolution"]
            samples(p, w, h, sc["renderer"]["spp"], g(name + "_samples.npz"))
        for name, (mk, kw) in list(scenes.INTEGRATE_CASES.items()) + list(scenes.OUTPUT_CASES.items()):
            integrate(mk(tmp, name=name + ".json", **kw), g(name + "_integrate.npz"))
        units(scenes.cornell(tmp, name="u_cornell.json", resolution=(96, 54), spp=1), g("cornell_units.json"))
        units(scenes.materialtest(tmp, name="u_materialtest.json", resolution=(96, 54), spp=1), g("materialtest_units.json"))
        for which in ("zoo_a", "zoo_b", "zoo_c", "zoo_d", "zoo_e", "zoo_f"):
            units(scenes.cornell_zoo(tmp, which, name="u_%s.json" % which, resolution=(96, 54), spp=1), g(which + "_units.json"))
        converged(scenes.cornell(tmp, name="c_cornell.json", resolution=(64, 36), spp=4096), 4096, g("cornell_converged.npz"), tmp)
        converged(scenes.materialtest(tmp, name="c_materialtest.json", resolution=(64, 36), spp=1024), 1024,
                  g("materialtest_converged.npz"), tmp)
        as_shipped_converged(tmp, g)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
