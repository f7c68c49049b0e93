#!/bin/bash
# round 6, session 55: k_tail (tail.hip) built -Os / without loop unrolling against the product's -O3: as shipped (a 2.7-4.6 ms k_tail launch per 16-spp pass); three times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s55; mkdir -p $O
i=0
for round in 1 2 3; do
  for v in prod tos tnounroll; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python tools/bench_as_shipped.py --repeats 4 > $O/s_$i.json 2> $O/s_$i.err
    python -c "
import json; s=json.loads(open('$O/s_$i.json').read().strip().splitlines()[-1]); print('%-10s'%'$v', s['msamples_per_s'])" 2>&1 | tail -1
    i=$((i+1))
  done
done
unset TUNGSTEN_AMD_LIB
TGHIP_VERBOSE=1 timeout 600 python tools/bench_as_shipped.py --repeats 2 2>&1 >/dev/null | grep "tail kernel" | tail -4
