#!/bin/bash
# round 6, session 59: the final library (traversal units and tail at -Os, shade_simple.hip max-ilp): the whole GPU suite, the driver's bench line, the other scenes, rocprofv3's kernel summary
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s59; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
for sc in cornell mesh1m instances10k; do
  timeout 700 python bench.py --scene $sc --no-cpu-baseline --no-extra > $O/bench_$sc.json 2> $O/bench_$sc.err
done
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_mt -o mt -- python $R/bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 2 > $R/$O/prof_mt.json 2> $R/$O/prof_mt.err )
find $O/prof_mt -name "*kernel_stats.csv" -exec cp {} $O/materialtest_kernel_stats.csv \;
find $O/prof_mt -type f -delete 2>/dev/null
timeout 600 python tools/bench_as_shipped.py --repeats 4 > $O/as_shipped.json 2> $O/as_shipped.err
timeout 900 python bench.py --scene instances10k --res 3840x2160 --spp 16 --steps 2 --no-cpu-baseline --no-extra --no-exclusive --count-spp 4 > $O/bench_c5_instances10k_4k_16spp.json 2> $O/bench_c5.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_s59/bench*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], d.get("value"), d.get("result_ok"), {k: v.get("value", v.get("msamples_per_s")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)},
              (d.get("sustained_clock") or {}).get("mhz_median"), r.get("frac"), (r.get("exclusive") or {}).get("frac"), r.get("traffic"))
    except Exception as e:
        print(f, "FAILED", str(e)[:80])
print(json.loads(open("gpurun_out/r6_s59/as_shipped.json").read().strip().splitlines()[-1])["msamples_per_s"])
PY
head -6 $O/materialtest_kernel_stats.csv | cut -c1-150
