"""Per-sample divergence of the device AND the oracle from the reference's goldens, and of the device from the oracle, case by case.

    python tools/device_vs_oracle.py [--out profiles/r4_device_diverge.jsonl] [case ...]          (needs a GPU)

For every golden case (tests/golden/*_samples.npz, rendered by the reference itself) three sets of (pixel, sample) are formed by the metric
of tests/test_oracle_golden.py (a sample agrees when every channel is within 1e-3 of the other's, relative to its largest channel):
    O = oracle  != reference      (coincident faces, Embree's rcp + Newton division: DESIGN.md section 8)
    D = device  != reference      (what tests/test_gpu_samples.py bounds)
    X = device  != oracle         (what the DEVICE adds: arithmetic that is not the host's, kernel bugs)
and the line reports |O|, |D|, |X|, |D \\ O| (device-only forks) and |O \\ D|, plus how many device samples equal the oracle's bit for bit.
A case whose X is empty has no device-specific divergence at all: its D is the oracle's O, whatever the cause of that is.
TEST INFRASTRUCTURE: the oracle is the checker here, never the thing measured."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def diverging(a, b):
    err = np.abs(a - b).max(axis=-1)
    return err > 1e-3*(np.abs(b).max(axis=-1) + 1e-3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "device_vs_oracle.jsonl"))
    ap.add_argument("cases", nargs="*")
    a = ap.parse_args()
    import oracle_lib
    import scenes
    import tungsten_amd as tg
    from test_oracle_golden import _oracle_samples
    names = a.cases or sorted(scenes.GOLDEN_CASES)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    tot = {"samples": 0, "O": 0, "D": 0, "X": 0}
    with open(a.out, "w") as f, tempfile.TemporaryDirectory() as tmp:
        for name in names:
            if name == "water_caustic" and not scenes.have_water_caustic():
                continue
            if ("materialtest" in name or name == "mesh1m") and not scenes.have_materialtest():
                continue
            mk, kw = scenes.GOLDEN_CASES[name]
            gold = np.load(os.path.join(scenes.GOLDEN, name + "_samples.npz"))
            ref, seed = gold["samples"], int(gold["seed"])
            h, w, spp, _ = ref.shape
            path = mk(tmp, name=name + ".json", **kw)
            r = tg.Renderer(path, seed=seed)
            sobol = bool(r.info.stratified_sampler)
            dev = r.trace_samples(0, spp, seed=seed, tile_seeds=oracle_lib.dice_tiles(w, h, seed)[0] if sobol else None)
            r.close()
            ora = _oracle_samples(mk, kw, name, tmp, ref, seed)
            O, D, X = diverging(ora, ref), diverging(dev, ref), diverging(dev, ora)
            line = {"case": name, "samples": int(O.size), "oracle_vs_ref": int(O.sum()), "device_vs_ref": int(D.sum()), "device_vs_oracle": int(X.sum()),
                    "device_only": int((D & ~O).sum()), "oracle_only": int((O & ~D).sum()),
                    "device_bit_equal_oracle": int((dev.view(np.uint32) == ora.view(np.uint32)).all(axis=-1).sum()),
                    "oracle_bit_equal_ref": int((ora.view(np.uint32) == ref.view(np.uint32)).all(axis=-1).sum()),
                    "device_bit_equal_ref": int((dev.view(np.uint32) == ref.view(np.uint32)).all(axis=-1).sum())}
            f.write(json.dumps(line) + "\n")
            f.flush()
            print(json.dumps(line))
            tot["samples"] += int(O.size); tot["O"] += int(O.sum()); tot["D"] += int(D.sum()); tot["X"] += int(X.sum())
    print("total: %d samples; oracle != reference %d, device != reference %d, device != oracle %d" % (tot["samples"], tot["O"], tot["D"], tot["X"]))


if __name__ == "__main__":
    main()
