#!/bin/bash
# round 6, session 12: the closest-hit walk's record test put off below N lanes with a record pending (PT_REC_MIN_LANES = 24 / 32 against 0)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s12; mkdir -p $O
L=$PWD/tungsten_amd/lib
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive"
for rep in 1 2; do
  for v in main rg24 rg32; do
    lib=$L/libtungsten_hip_$v.so; [ $v = main ] && lib=$L/libtungsten_hip.so
    TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --steps 6 > $O/ab_materialtest_${v}_$rep.json 2>> $O/ab.err
  done
done
for v in main rg24 rg32; do
  lib=$L/libtungsten_hip_$v.so; [ $v = main ] && lib=$L/libtungsten_hip.so
  TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --scene mesh1m > $O/ab_mesh1m_$v.json 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_s12/ab_*.json")):
    try:
        d = json.load(open(f)); k = d.get("kernels", {})
        w = (d.get("walk") or {}).get("closest_hit", {})
        print("%-36s %8.2f Msamples/s ok %s %s busy %s turns/ray %s rec %s node %s nodes/ray %s" % (os.path.basename(f), d["value"], d["result_ok"], {n: v["avg_us"] for n, v in k.items()}, w.get("busy_lanes_per_turn"), w.get("turns_per_ray"), w.get("record_test"), w.get("node_visit"), w.get("nodes_per_ray")))
    except Exception as e:
        print(f, "FAILED", e)
PY
cat $O/ab_summary.txt
