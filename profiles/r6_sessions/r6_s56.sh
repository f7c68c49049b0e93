#!/bin/bash
# round 6, session 56: the library with tail.hip at -Os: the whole GPU suite, as shipped, the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s56; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
timeout 600 python tools/bench_as_shipped.py --repeats 4 > $O/as_shipped.json 2> $O/as_shipped.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_s56/bench_default.json"))
r = d["roofline"]
print(d["value"], d["result_ok"], {k: v.get("value", v.get("msamples_per_s")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)}, (d.get("sustained_clock") or {}).get("mhz_median"))
print("roofline", r.get("kernel"), r.get("frac"), "traffic", r.get("traffic"), "exclusive", (r.get("exclusive") or {}).get("frac"), "lanes", (r["valu"].get("lane_utilisation") or {}).get("loop"))
print(json.loads(open("gpurun_out/r6_s56/as_shipped.json").read().strip().splitlines()[-1])["msamples_per_s"])
PY
