"""Effective shader clock per kernel class: GRBM_GUI_ACTIVE (cycles the graphics engine was active during a dispatch) over the dispatch's
wall time, as MI355X_MICROARCH.md "DVFS give-back" prescribes -- what bench.py's VALU line should be priced at instead of the 2.4 GHz maximum.

    python tools/pmc_clock.py --out profiles/r6_clock.json [--scene materialtest] [--spp 32]

One `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` child pass of bench.py (counter collection serialises the dispatches: every launch has the
chip to itself, and profiled passes clock a little lower than unprofiled ones -- the guide measured 1.89-1.95 against 2.02 GHz -- so this is a
lower bound of the timed region's clock; `rocm-smi --showclocks` sampled during an unprofiled run is the other reading, tools/session)."""
import argparse
import csv
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--scene", default="materialtest")
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="pmcclk_")
    out = os.path.join(tmp, "o")
    cmd = [shutil.which("rocprofv3"), "--pmc", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "clk", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--scene", a.scene, "--spp", str(a.spp), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-extra", "--no-kernel-timing", "--no-traffic"] + [x for kv in a.opt for x in ("--opt", kv)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=600)
    files = {os.path.basename(f): os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".csv")}
    cc = [v for k, v in files.items() if k.endswith("counter_collection.csv")]
    kt = [v for k, v in files.items() if k.endswith("kernel_trace.csv")]
    if p.returncode != 0 or not cc:
        sys.stderr.write(p.stdout[-1500:])
        raise SystemExit("rocprofv3 pass failed (rc %d)" % p.returncode)
    span = {}
    if kt:
        with open(kt[0]) as f:
            for row in csv.DictReader(f):
                span[row.get("Dispatch_Id")] = (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))
    per = {}
    with open(cc[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != "GRBM_GUI_ACTIVE":
                continue
            k = re.sub(r"<.*", "", re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip())
            if not k.startswith("k_"):
                continue
            t = span.get(row.get("Dispatch_Id"))
            if t is None and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                t = (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))
            if t is None or t[1] <= t[0]:
                continue
            e = per.setdefault(k, {"cycles": 0.0, "ns": 0, "launches": 0})
            # (GRBM_GUI_ACTIVE is reported once per XCD / shader engine instance: rows of one dispatch are averaged by taking the maximum)
            d = e.setdefault("_d", {})
            key = row.get("Dispatch_Id")
            if key not in d:
                d[key] = [0.0, t[1] - t[0]]
            d[key][0] = max(d[key][0], float(row["Counter_Value"]))
    res = {}
    for k, e in sorted(per.items()):
        cyc = sum(v[0] for v in e["_d"].values())
        ns = sum(v[1] for v in e["_d"].values())
        res[k] = {"launches": len(e["_d"]), "gui_active_cycles": round(cyc), "dispatch_ns": ns, "effective_clock_ghz": round(cyc/ns, 4) if ns else None}
    tot_c = sum(r["gui_active_cycles"] for r in res.values())
    tot_n = sum(r["dispatch_ns"] for r in res.values())
    data = json.load(open(a.out)) if os.path.exists(a.out) else {}
    data["%s@%dspp" % (a.scene, a.spp)] = {"per_kernel": res, "all_kernels_effective_clock_ghz": round(tot_c/tot_n, 4) if tot_n else None,
                                         "note": "GRBM_GUI_ACTIVE summed per dispatch (max over the counter's instances) / dispatch duration from the kernel trace"}
    json.dump(data, open(a.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(data, indent=1, sort_keys=True))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
