#!/bin/bash
# round 6, session 5: the three lane-cliff changes one at a time (session 4: together they cost the shadow walk 640 -> 790 us), the fixed second-ray
# preparation against the adaptive suite, and the instanced walk's section counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s5; mkdir -p $O
L=$PWD/tungsten_amd/lib
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive"
for rep in 1 2; do
  for v in r6base tri prep late all; do
    lib=$L/libtungsten_hip_$v.so; [ $v = all ] && lib=$L/libtungsten_hip.so
    TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --steps 6 > $O/ab_materialtest_${v}_$rep.json 2>> $O/ab.err
  done
done
for v in r6base tri prep late all; do
  lib=$L/libtungsten_hip_$v.so; [ $v = all ] && lib=$L/libtungsten_hip.so
  TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --scene mesh1m > $O/ab_mesh1m_$v.json 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_s5/ab_*.json")):
    try:
        d = json.load(open(f)); k = d.get("kernels", {})
        print("%-40s %8.2f Msamples/s  %s" % (os.path.basename(f), d["value"], {n: v["avg_us"] for n, v in k.items()}))
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 600 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_samples.py -m gpu -q -x > $O/gpu_adaptive_samples.txt 2>&1
tail -3 $O/gpu_adaptive_samples.txt
timeout 600 python bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic > $O/bench_instances10k.json 2> $O/bench_instances10k.err
cat $O/ab_summary.txt
