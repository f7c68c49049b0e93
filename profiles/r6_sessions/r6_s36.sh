#!/bin/bash
# round 6, session 36: evidence on the final binaries (atmospheric medium, restated double libm, media helpers, the host's turn) -- the whole GPU suite, the driver's bench
# line (with sustained_clock), the other scenes, rocprofv3's kernel summary of the metric's workload, as shipped, the media scenes, configs[4]'s scene at 4K
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s36; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
/usr/bin/time -v -o $O/bench_default.time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
grep "Elapsed" $O/bench_default.time
for sc in cornell mesh1m instances10k; do
  timeout 700 python bench.py --scene $sc --no-cpu-baseline --no-extra > $O/bench_$sc.json 2> $O/bench_$sc.err
done
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_mt -o mt -- python $R/bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 2 > $R/$O/prof_mt.json 2> $R/$O/prof_mt.err )
find $O/prof_mt -name "*kernel_stats.csv" -exec cp {} $O/materialtest_kernel_stats.csv \;
find $O/prof_mt -type f -delete 2>/dev/null
timeout 600 python tools/bench_as_shipped.py --repeats 4 > $O/as_shipped.json 2> $O/as_shipped.err
timeout 600 python tools/bench_media.py 64 > $O/media.jsonl 2> $O/media.err
timeout 900 python bench.py --scene instances10k --res 3840x2160 --spp 16 --steps 2 --no-cpu-baseline --no-extra --no-exclusive --count-spp 4 > $O/bench_c5_instances10k_4k_16spp.json 2> $O/bench_c5.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_s36/bench*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d.get("value"), d.get("result_ok"), {k: v.get("value") for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)}, d.get("sustained_clock", {}) and d["sustained_clock"].get("mhz_median"),
              d["roofline"].get("frac"), (d["roofline"].get("exclusive") or {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", str(e)[:80])
print(open("gpurun_out/r6_s36/as_shipped.json").read().strip().splitlines()[-1][:200])
for l in open("gpurun_out/r6_s36/media.jsonl"):
    d = json.loads(l); print(d["scene"], round(d["msamples_per_s"]))
PY
head -8 $O/materialtest_kernel_stats.csv | cut -c1-160
