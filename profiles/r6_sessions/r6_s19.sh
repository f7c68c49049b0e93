#!/bin/bash
# round 6, session 19: the one-launch render (Cornell box) tracing a slot's next ray at the end of the turn (PT_TRACE_AHEAD) -- the GPU suite, then the A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s19; mkdir -p $O
L=$PWD/tungsten_amd/lib
timeout 1800 python -m pytest tests -m gpu -q -x > $O/gpu_suite.txt 2>&1
grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive"
for rep in 1 2; do
  for v in nota main; do
    lib=$L/libtungsten_hip_$v.so; [ $v = main ] && lib=$L/libtungsten_hip.so
    TUNGSTEN_AMD_LIB=$lib timeout 300 python bench.py $Q --scene cornell --steps 3 > $O/ab_cornell_${v}_$rep.json 2>> $O/ab.err
  done
done
for v in nota main; do
  lib=$L/libtungsten_hip_$v.so; [ $v = main ] && lib=$L/libtungsten_hip.so
  TUNGSTEN_AMD_LIB=$lib timeout 300 python tools/bench_as_shipped.py --scene cornell --repeats 3 > $O/as_shipped_cornell_$v.json 2>> $O/ab.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_s19/*.json")):
    try:
        d = json.load(open(f))
        print("%-34s %8.2f Msamples/s ok %s mean %s rays/sample %s" % (os.path.basename(f), d.get("value", d.get("msamples_per_s")), d.get("result_ok"), d.get("image_mean"), d.get("rays_per_sample", d.get("closest_rays_per_sample"))))
    except Exception as e:
        print(f, "FAILED", e)
PY
