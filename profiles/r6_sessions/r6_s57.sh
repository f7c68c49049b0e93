#!/bin/bash
# round 6, session 57: the traversal units at -Os (wos: both; wsos: walk_shadow.hip only) against -O3: metric, mesh1m, instances10k; twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s57; mkdir -p $O
i=0
for round in 1 2; do
  for v in prod wos wsos; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 10 > $O/mt_$i.json 2> $O/mt_$i.err
    timeout 600 python bench.py --scene mesh1m --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 4 > $O/mesh_$i.json 2> $O/mesh_$i.err
    timeout 600 python bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 3 > $O/inst_$i.json 2> $O/inst_$i.err
    python -c "
import json
m=json.load(open('$O/mt_$i.json')); b=json.load(open('$O/mesh_$i.json')); a=json.load(open('$O/inst_$i.json'))
print('%-5s'%'$v', 'materialtest', m['value'], {k: round(x['avg_us']) for k, x in m['kernels'].items()}, m['image_mean'][0], '| mesh1m', b['value'], '| instances10k', a['value'])" 2>&1 | tail -1
    i=$((i+1))
  done
done
