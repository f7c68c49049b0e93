#!/usr/bin/env python3
"""A/B sweeps of shim options on ONE box in ONE process (boxes of the pool differ by up to 10 %, so variants are only
comparable within a run): every option set renders the same workload through bench.py's timed loop.

    python tools/sweep.py --scene materialtest --spp 256 --steps 3 -- "suspend_lanes=0" "suspend_lanes=16,suspend_turns=32" ...

One JSON line per option set: Msamples/s, ms per step, per-launch averages of the three kernel classes, iterations."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="materialtest")
    ap.add_argument("--material", default="shipped")
    ap.add_argument("--res", default="")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=1, help="go through the option sets this many times (A B C A B C)")
    ap.add_argument("--emulate-shards", type=int, default=0)
    ap.add_argument("sets", nargs="*", help="comma-separated key=value lists; '-' = defaults")
    a = ap.parse_args()
    w, h = (int(v) for v in (a.res or ("1280x720" if a.scene in ("materialtest", "cornell") else "1920x1080")).split("x"))
    spp = a.spp or (256 if a.scene in ("materialtest", "cornell") else 32)
    ba = argparse.Namespace(gpus=1, steps=a.steps, warmup=a.warmup, scene=a.scene, material=a.material, res="%dx%d" % (w, h), spp=spp,
                            no_cpu_baseline=True, no_extra=True, no_kernel_timing=False, cpu_seconds=0.0, traffic=False, opt=[],
                            emulate_shards=a.emulate_shards, count_spp=0, exclusive=False)
    b = bench.Bench(ba)
    b.shared_ctx = b.tg.lib.tghip_create(0)      # one context for every option set (options persist: name every swept key in every set)
    try:
        for rep in range(a.repeat):
            for s in (a.sets or ["-"]):
                ba.opt = [] if s == "-" else s.split(",")
                r = b.run(a.scene, w, h, spp, a.steps, a.warmup, False)
                k = r["kernels"]
                line = {"opts": s, "value": r["value"], "ms_per_step": r["ms_per_step"], "iterations": r["wavefront_iterations"],
                        "us": {n: k[n]["avg_us"] for n in k}, "ok": r["result_ok"], "mean": r["image_mean"]}
                if "emulated_shards" in r:
                    line["emulated_shards"] = r["emulated_shards"]
                print(json.dumps(line), flush=True)
    finally:
        b.close()


if __name__ == "__main__":
    main()
