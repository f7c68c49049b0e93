"""Hardware counters per kernel VARIANT (template arguments kept), one `rocprofv3 --pmc` child pass per counter group.

    python tools/pmc_variants.py --out profiles/r5_sq_counters.json [--scene materialtest] [--spp 32] [--opt key=value ...] [--groups sq,ifetch,tcp,sqc]

Each group is one pass of `bench.py --steps 1 --warmup 0` under the profiler (counter collection serialises the dispatches, so every
launch has the chip to itself: these are exclusive figures, unlike the four-part schedule of a timed run).  Per variant the totals of
every counter over all its dispatches, the dispatch count, and the derived fractions MI355X_MICROARCH.md "rocprofv3 PMC slots" names:
SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing) ~= SQ_WAVE_CYCLES."""
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
GROUPS = {
    "sq": ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES"],
    "ifetch": ["SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"],
    "mem": ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS"],
    "sqc": ["SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQC_DCACHE_MISSES_DUPLICATE"],
    "tcp": ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"],
    # enabled lanes per issued VALU instruction (counter_defs.yaml: VALUUtilization = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64); the pair is
    # calibrated by tools/ubench_lanes.hip, profiles/r6_ubench_lanes.txt)
    "lane": ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_SALU", "SQ_WAIT_ANY"],
}


def variant(name):
    k = re.sub(r"\(.*", "", name).replace("void ", "").strip()
    return k


def one_pass(counters, a, tmp, tag):
    exe = shutil.which("rocprofv3")
    out = os.path.join(tmp, "pmc_" + tag)
    cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--scene", a.scene, "--spp", str(a.spp), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-extra", "--no-kernel-timing", "--no-traffic"] + [x for kv in a.opt for x in ("--opt", kv)]
    if a.res:
        cmd += ["--res", a.res]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=a.timeout)
    except subprocess.TimeoutExpired:
        sys.stderr.write("pass %s timed out after %d s\n" % (tag, a.timeout))
        return {}
    files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
    if p.returncode != 0 or not files:
        sys.stderr.write("pass %s failed (rc %d): %s\n" % (tag, p.returncode, p.stdout[-800:]))
        return {}
    sums = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            k = variant(row["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            e = sums.setdefault(k, {"_ids": set()})
            e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            e["_ids"].add(row.get("Dispatch_Id"))
    shutil.rmtree(out, ignore_errors=True)
    return sums


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--scene", default="materialtest")
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--res", default="")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--groups", default="sq,ifetch,mem,sqc,tcp")
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="pmcv_")
    data = json.load(open(a.out)) if os.path.exists(a.out) else {}
    key = a.label or ("%s@%dspp%s" % (a.scene, a.spp, ("," + ",".join(a.opt)) if a.opt else ""))
    run = data.setdefault(key, {})
    for g in a.groups.split(","):
        sums = one_pass(GROUPS[g], a, tmp, g)
        for k, e in sums.items():
            r = run.setdefault(k, {})
            r["launches"] = len(e.pop("_ids"))
            for n, v in e.items():
                r[n] = round(v)
    for k, r in sorted(run.items()):
        wave = r.get("SQ_WAVE_CYCLES", 0)
        if wave:
            for n, label in (("SQ_WAIT_ANY", "frac_waiting"), ("SQ_WAIT_INST_ANY", "frac_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "frac_issuing"),
                             ("SQ_ACTIVE_INST_VALU", "frac_issuing_valu")):
                if n in r:
                    r[label] = round(r[n]/float(wave), 4)
        if r.get("SQ_THREAD_CYCLES_VALU") and r.get("SQ_ACTIVE_INST_VALU"):
            r["lane_utilisation"] = round(r["SQ_THREAD_CYCLES_VALU"]/(64.0*r["SQ_ACTIVE_INST_VALU"]), 4)
        if r.get("SQC_ICACHE_REQ"):
            r["icache_hit_rate"] = round(r.get("SQC_ICACHE_HITS", 0)/float(r["SQC_ICACHE_REQ"]), 4)
        if r.get("SQC_DCACHE_REQ"):
            r["scalar_dcache_hit_rate"] = round(r.get("SQC_DCACHE_HITS", 0)/float(r["SQC_DCACHE_REQ"]), 4)
        if r.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            r["l1_hit_rate"] = round(1.0 - r.get("TCP_TCC_READ_REQ_sum", 0)/float(r["TCP_TOTAL_CACHE_ACCESSES_sum"]), 4)
        if r.get("TCP_TCC_READ_REQ_sum"):
            r["l1_miss_latency_cycles"] = round(r.get("TCP_TCC_READ_REQ_LATENCY_sum", 0)/float(r["TCP_TCC_READ_REQ_sum"]), 1)
        if r.get("TCC_REQ_sum"):
            r["l2_hit_rate"] = round(r.get("TCC_HIT_sum", 0)/float(r["TCC_REQ_sum"]), 4)
        print("%-70s %s" % (k[:70], {n: v for n, v in r.items() if n.startswith("frac") or n.endswith("rate") or n == "lane_utilisation" or n.endswith("cycles") and n.startswith("l1")}))
    json.dump(data, open(a.out, "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
