#!/bin/bash
# round 6, session 61: the traversal units at -Oz / -O1 / -Os without loop unrolling against the product's -Os; metric's workload and mesh1m, three alternations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s61; mkdir -p $O
i=0
for round in 1 2 3; do
  for v in prod woz wo1 wosnu; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 12 > $O/mt_$i.json 2> $O/mt_$i.err
    timeout 600 python bench.py --scene mesh1m --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 5 > $O/mesh_$i.json 2> $O/mesh_$i.err
    python -c "
import json
m=json.load(open('$O/mt_$i.json')); b=json.load(open('$O/mesh_$i.json'))
print('%-6s'%'$v', 'materialtest', m['value'], {k: round(x['avg_us']) for k, x in m['kernels'].items()}, m['image_mean'][0], '| mesh1m', b['value'])" 2>&1 | tail -1
    i=$((i+1))
  done
done
