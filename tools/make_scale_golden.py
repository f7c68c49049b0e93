"""TEST INFRASTRUCTURE (build container only: needs oracle/_ref/ref_harness).  The reference's own per-sample radiance
(PathTracer::traceSample driven by oracle/ref_harness.cpp, the shared counter-based random stream) for a handful of golden cases at EIGHT
times the goldens' samples -- twice the resolution in x and y, twice the samples per pixel --, kept as one 16-bit hash per sample
(tests/golden/scale8_<case>.npz, scale64_<case>.npz: float32 radiance of 200 000-330 000 samples would be megabytes per case; a hash answers the only question
asked of it -- is the device's sample the reference's, bit for bit -- and still counts the samples that are not):

    python tools/make_scale_golden.py [case ...]

tests/test_gpu_scale.py renders the same samples on the device and compares hash by hash."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import tungsten_amd as tg  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
SEED = tg.DEFAULT_SEED
# the scenes the stress renders of round 4 found differing samples in (cornell_bump, mesh1m), the metric's scene, the instanced one, and the
# flat lists made of coincident faces whose order the top-level tree decides
CASES = ["cornell_bump", "mesh1m", "materialtest", "cornell_instances", "cornell_ties", "cornell_round_ties", "cornell_crowd"]
# (resolution factor, spp factor): 8 times the goldens' samples for every case, 64 times for the three in which round 4's stress renders
# of the ORACLE found samples that are not the reference's (profiles/r4_oracle_stress_64x_all.txt: cornell_bump, mesh1m, materialtest_sobol)
# (round 6: cornell_atmosphere at 64 times -- the one place the device's arithmetic is not the host's: AtmosphericMedium::inverseOpticalDepth's double-precision
# erf / exp / log are ocml's on the device, glibc's in the reference)
SIZES = {"scale8": (2, 2, CASES), "scale64": (4, 4, ["cornell_bump", "mesh1m", "materialtest_sobol", "cornell_atmosphere"])}


def sample_hash(a):
    """uint16 per sample of a float32 [..., 3] radiance array: a multiplicative mix of the three bit patterns (equal bits <=> equal hash, up to 2^-16)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    h = (u[..., 0]*np.uint64(0x9E3779B1) + u[..., 1]*np.uint64(0x85EBCA77) + u[..., 2]*np.uint64(0xC2B2AE3D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h*np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    return (h & np.uint64(0xFFFF)).astype(np.uint16)


def scaled_case(name, tmp, size="scale8"):
    mk, kw = scenes.GOLDEN_CASES[name]
    w0, h0 = kw["resolution"]
    scale, factor, _ = SIZES[size]
    kw = dict(kw, resolution=(w0*scale, h0*scale), spp=kw["spp"]*factor)
    return mk(tmp, name="%s_%s.json" % (name, size), **kw), kw


def main():
  for size in ("scale8", "scale64"):
    for name in SIZES[size][2]:
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        tmp = tempfile.mkdtemp(prefix="tg_scale_")
        path, kw = scaled_case(name, tmp, size)
        (w, h), spp = kw["resolution"], kw["spp"]
        out = os.path.join(tmp, "s.bin")
        subprocess.check_call([HARNESS, "samples", path, str(SEED), str(spp), out], stdout=subprocess.DEVNULL, cwd=os.path.dirname(path))
        ref = np.fromfile(out, np.float32).reshape(h, w, spp, 3)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "%s_%s.npz" % (size, name)), hash=sample_hash(ref), seed=np.uint32(SEED),
                            mean=ref.mean(axis=(0, 1, 2), dtype=np.float64), finite=np.uint64(np.isfinite(ref).all(axis=-1).sum()))
        print("%-8s %-24s %dx%d @ %d spp = %d samples, mean %s" % (size, name, w, h, spp, h*w*spp, ref.mean(axis=(0, 1, 2))), flush=True)


if __name__ == "__main__":
    main()
