#!/bin/bash
# round 6, session 37: `python bench.py` exactly as the driver runs it, on the final binaries (session 36's copy of this step tripped over a missing /usr/bin/time)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s37; mkdir -p $O
T0=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
T1=$(date +%s.%N)
echo "bench.py wall: $(echo "$T1 - $T0" | bc) s" | tee $O/bench_default.wall
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_s37/bench_default.json"))
print(d["value"], d["result_ok"], {k: v.get("value", v.get("msamples_per_s")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)}, d.get("sustained_clock"))
r = d["roofline"]
print("roofline", r.get("kernel"), r.get("frac"), "traffic", r.get("traffic"), "exclusive", {k: (r.get("exclusive") or {}).get(k) for k in ("kernel", "frac", "achieved")})
print("valu", {k: r["valu"].get(k) for k in ("frac", "frac_at_sustained_clock")}, (r["valu"].get("lane_utilisation") or {}).get("loop"), ((r["valu"].get("priced") or {}).get("static_mix") or {}).get("frac_of_simd_time_at_sustained_clock"))
print("cpu_baseline", d.get("cpu_baseline"))
PY
