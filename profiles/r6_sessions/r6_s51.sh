#!/bin/bash
# round 6, session 51: the Cornell box's kernel (shade_simple.hip, max-ilp) with the SLP vectoriser back on / without loop unrolling / at -O2; Cornell box 1280x720x256, the list twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s51; mkdir -p $O
i=0
for round in 1 2; do
  for v in prod cslp cnounroll co2ilp; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --scene cornell --no-cpu-baseline --no-extra --no-traffic --no-clock --steps 6 > $O/b_$i.json 2> $O/b_$i.err
    python -c "
import json; d=json.load(open('$O/b_$i.json')); print('%-10s'%'$v', d['value'], d['image_mean'])" 2>&1 | tail -1
    i=$((i+1))
  done
done
