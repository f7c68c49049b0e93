#!/bin/bash
# round 6, session 18: evidence on the final binaries (after k_trace_shadow_fast_inst) -- the whole GPU suite, the driver's bench line, the other scenes, rocprofv3 stats of instances10k
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s18; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
for sc in cornell mesh1m instances10k; do
  timeout 700 python bench.py --scene $sc --no-cpu-baseline --no-extra > $O/bench_$sc.json 2> $O/bench_$sc.err
done
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_inst -o inst -- python $R/bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --no-exclusive --steps 2 > $R/$O/prof_inst.json 2> $R/$O/prof_inst.err )
find $O/prof_inst -name "*kernel_stats.csv" -exec cp {} $O/instances10k_kernel_stats.csv \;
find $O/prof_inst -type f -delete 2>/dev/null
timeout 900 python tools/pmc_variants.py --out $O/sq_counters_instances10k.json --scene instances10k --spp 32 --groups lane,sq,tcp,tcc --timeout 150 > $O/sq_counters_instances10k.txt 2>&1
timeout 900 python bench.py --scene instances10k --res 3840x2160 --spp 16 --steps 2 --no-cpu-baseline --no-extra --no-exclusive --count-spp 4 > $O/bench_c5_instances10k_4k_16spp.json 2> $O/bench_c5.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_s18/bench*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d.get("value"), d.get("result_ok"), {k: v.get("value") for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)}, list(d.keys())[:14], list(d["roofline"].keys())[:12])
    except Exception as e:
        print(f, "FAILED", str(e)[:80])
PY
