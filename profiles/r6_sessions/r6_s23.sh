#!/bin/bash
# round 6, session 23: counters of the metric's workload on the final binaries; wall time of the driver's bench command; smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s23; mkdir -p $O
timeout 900 python tools/pmc_variants.py --out $O/sq_counters_materialtest.json --scene materialtest --spp 32 --groups lane,sq,tcp,tcc --timeout 150 > $O/sq_counters_materialtest.txt 2>&1
/usr/bin/time -v python bench.py > $O/bench_default.json 2> $O/bench_default.err; grep "Elapsed" $O/bench_default.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_s23/bench_default.json")); print(d["value"], d["result_ok"], {k: v.get("value") for k, v in d["extra"].items()})
c = json.load(open("gpurun_out/r6_s23/sq_counters_materialtest.json")); c = list(c.values())[0]
for k, v in c.items():
    if isinstance(v, dict) and "SQ_WAVE_CYCLES" in v: print("%-52s issuing %.2f valu %.2f waiting %.2f l1 %.2f l2 %.2f lanes %.2f" % (k[:52], v["frac_issuing"], v["frac_issuing_valu"], v["frac_waiting"], v["l1_hit_rate"], v["l2_hit_rate"], v["lane_utilisation"]))
PY
