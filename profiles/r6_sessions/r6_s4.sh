#!/bin/bash
# round 6, session 4: A/B of the lane-cliff changes (branchless triangle test, shadow ray prepared at the refill, late publishes), GPU suite, k_shade sections
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s4; mkdir -p $O
B=$PWD/tungsten_amd/lib/libtungsten_hip_r6base.so
Q="--no-cpu-baseline --no-extra --no-traffic"
for rep in 1 2; do
  TUNGSTEN_AMD_LIB=$B timeout 300 python bench.py $Q --steps 6 > $O/ab_materialtest_base_$rep.json 2>> $O/ab.err
  timeout 300 python bench.py $Q --steps 6 > $O/ab_materialtest_new_$rep.json 2>> $O/ab.err
done
for sc in mesh1m instances10k cornell; do
  TUNGSTEN_AMD_LIB=$B timeout 300 python bench.py $Q --scene $sc > $O/ab_${sc}_base.json 2>> $O/ab.err
  timeout 300 python bench.py $Q --scene $sc > $O/ab_${sc}_new.json 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json, glob, os
for f in sorted(glob.glob(os.environ.get("O", "gpurun_out/r6_s4") + "/ab_*.json")):
    try:
        d = json.load(open(f)); k = d.get("kernels", {})
        print("%-40s %8.2f Msamples/s  %s" % (os.path.basename(f), d["value"], {n: v["avg_us"] for n, v in k.items()}))
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_suite.txt 2>&1
tail -3 $O/gpu_suite.txt
P=$PWD/tungsten_amd/lib/libtungsten_hip_prof.so
TGHIP_VERBOSE=1 TUNGSTEN_AMD_LIB=$P timeout 300 python bench.py $Q --no-kernel-timing --scene cornell --spp 64 --steps 1 --warmup 0 > $O/prof_cornell.json 2> $O/prof_cornell.txt
TGHIP_VERBOSE=1 TUNGSTEN_AMD_LIB=$P timeout 300 python bench.py $Q --no-kernel-timing --spp 32 --steps 1 --warmup 0 > $O/prof_materialtest.json 2> $O/prof_materialtest.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default_new.json 2> $O/bench_default_new.err
cat $O/ab_summary.txt
