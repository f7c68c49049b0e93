#!/bin/bash
# round 6, session 35: the host's turn between two adaptive passes -- errorPercentile95 by counting instead of nth_element, the stochastic rounding without a branch,
# the record passes' hint table filled by a launch instead of built and uploaded by the host -- parity of every record of every pass, then as shipped against the previous commit's library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s35; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_outputs.py tests/test_ref_binding.py -m gpu -q > $O/gpu_adaptive.txt 2>&1
tail -4 $O/gpu_adaptive.txt
for round in 1 2 3; do
  TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_base.so timeout 600 python tools/bench_as_shipped.py --repeats 4 > $O/shipped_base_$round.json 2> $O/shipped_base_$round.err
  timeout 600 python tools/bench_as_shipped.py --repeats 4 > $O/shipped_new_$round.json 2> $O/shipped_new_$round.err
done
TGHIP_VERBOSE=1 timeout 600 python tools/bench_as_shipped.py --repeats 2 > /dev/null 2> $O/shipped_verbose.txt
grep "pass spp" $O/shipped_verbose.txt | tail -4 | cut -c1-260
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_s35/shipped_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d.get("msamples_per_s"), d.get("seconds"))
    except Exception as e:
        print(f, "FAILED", e)
PY
