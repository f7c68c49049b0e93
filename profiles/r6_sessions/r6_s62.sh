#!/bin/bash
# round 6, session 62: walk_shadow.hip at -O1 (its launches read 602 -> 590 us in session 61) with tungsten_hip.hip at the product's -Os; metric, mesh1m, instances10k; five alternations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s62; mkdir -p $O
i=0
for round in 1 2 3 4 5; do
  for v in prod wso1; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 12 > $O/mt_$i.json 2> $O/mt_$i.err
    timeout 600 python bench.py --scene mesh1m --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 5 > $O/mesh_$i.json 2> $O/mesh_$i.err
    python -c "
import json
m=json.load(open('$O/mt_$i.json')); b=json.load(open('$O/mesh_$i.json'))
print('%-5s'%'$v', 'materialtest', m['value'], {k: round(x['avg_us']) for k, x in m['kernels'].items()}, m['image_mean'][0], '| mesh1m', b['value'])" 2>&1 | tail -1
    i=$((i+1))
  done
done
