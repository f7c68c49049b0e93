#!/bin/bash
# round 6, session 21: the flat list's record loop with the next record's rows requested ahead (PT_FLAT_PIPELINE) -- flat-scene parity, then the A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s21; mkdir -p $O
L=$PWD/tungsten_amd/lib
timeout 1800 python -m pytest tests/test_gpu_samples.py tests/test_gpu_parity.py tests/test_gpu_adaptive.py tests/test_gpu_outputs.py -m gpu -q -x > $O/gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive"
for rep in 1 2; do
  for v in nofp main; do
    lib=$L/libtungsten_hip_$v.so; [ $v = main ] && lib=$L/libtungsten_hip.so
    TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --scene cornell --steps 3 > $O/ab_cornell_${v}_$rep.json 2>> $O/ab.err
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_s21/*.json")):
    try:
        d = json.load(open(f))
        print("%-34s %8.2f Msamples/s ok %s mean %s" % (os.path.basename(f), d.get("value"), d.get("result_ok"), d.get("image_mean")))
    except Exception as e:
        print(f, "FAILED", e)
PY
