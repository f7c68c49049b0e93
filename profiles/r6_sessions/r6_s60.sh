#!/bin/bash
# round 6, session 60: shade_class.hip (the conductor family's shading, a fifth of the metric's kernel time) at -Os / -O2 against -O3 on the final library; metric's workload, four alternations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s60; mkdir -p $O
i=0
for round in 1 2 3 4; do
  for v in prod clos clo2; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 12 > $O/mt_$i.json 2> $O/mt_$i.err
    python -c "
import json
m=json.load(open('$O/mt_$i.json'))
print('%-5s'%'$v', 'materialtest', m['value'], {k: round(x['avg_us']) for k, x in m['kernels'].items()}, m['image_mean'][0])" 2>&1 | tail -1
    i=$((i+1))
  done
done
