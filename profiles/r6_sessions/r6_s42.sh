#!/bin/bash
# round 6, session 42: the whole GPU suite and the driver's bench line on the round's final library (equirectangular + cubemap cameras through k_camera_rays, the finish_lean variant behind
# its option: the default launches' kernels are instruction for instruction those of sessions 36 / 37)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s42; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_s42/bench_default.json"))
r = d["roofline"]
print(d["value"], d["result_ok"], {k: v.get("value", v.get("msamples_per_s")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)}, (d.get("sustained_clock") or {}).get("mhz_median"))
print("roofline", r.get("kernel"), r.get("frac"), "traffic", r.get("traffic"), "exclusive", (r.get("exclusive") or {}).get("frac"), "lanes", (r["valu"].get("lane_utilisation") or {}).get("loop"))
PY
