#!/bin/bash
# round 6, session 46: the Cornell box's one-launch kernel (shade_simple.hip) under other compiler options -- -O2, -Os, the AMDGPU scheduler strategies max-ilp / max-memory-clause -- against
# the product's -O3; Cornell box 1280x720x256, the list twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s46; mkdir -p $O
i=0
for round in 1 2; do
  for v in prod o2 os ilp mem; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_ss$v.so; fi
    timeout 600 python bench.py --scene cornell --no-cpu-baseline --no-extra --no-traffic --no-clock --steps 6 > $O/b_$i.json 2> $O/b_$i.err
    python -c "
import json; d=json.load(open('$O/b_$i.json')); print('%-5s'%'$v', d['value'], d['image_mean'])" 2>&1 | tail -1
    i=$((i+1))
  done
done
