#!/bin/bash
# round 6, session 11: the closest-hit walk with a ray in store per lane (PT_PREFETCH) -- parity suites, then the A/B against the refill at <= 40 busy lanes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s11; mkdir -p $O
L=$PWD/tungsten_amd/lib
timeout 1200 python -m pytest tests/test_gpu_samples.py tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x > $O/gpu_parity.txt 2>&1
tail -4 $O/gpu_parity.txt
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive"
for rep in 1 2; do
  for v in nopf main pf32 pf48 pfw4; do
    lib=$L/libtungsten_hip_$v.so; [ $v = main ] && lib=$L/libtungsten_hip.so
    [ -f $lib ] || continue
    TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --steps 6 > $O/ab_materialtest_${v}_$rep.json 2>> $O/ab.err
  done
done
for v in nopf main pf32 pf48 pfw4; do
  lib=$L/libtungsten_hip_$v.so; [ $v = main ] && lib=$L/libtungsten_hip.so
  [ -f $lib ] || continue
  TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --scene mesh1m > $O/ab_mesh1m_$v.json 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_s11/ab_*.json")):
    try:
        d = json.load(open(f)); k = d.get("kernels", {})
        w = (d.get("walk") or {}).get("closest_hit", {})
        print("%-36s %8.2f Msamples/s  %s  busy %s turns/ray %s refill %s" % (os.path.basename(f), d["value"], {n: v["avg_us"] for n, v in k.items()}, w.get("busy_lanes_per_turn"), w.get("turns_per_ray"), w.get("refill")))
    except Exception as e:
        print(f, "FAILED", e)
PY
cat $O/ab_summary.txt
