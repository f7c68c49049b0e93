#!/bin/bash
# round 6, session 52: max-ilp on the traversal translation units (tungsten_hip.hip: the closest-hit walks incl. round 6's k_trace_closest_instw; walk_shadow.hip) -- instances10k, the metric, mesh1m; twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s52; mkdir -p $O
i=0
for round in 1 2; do
  for v in prod wilp; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 3 > $O/inst_$i.json 2> $O/inst_$i.err
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 10 > $O/mt_$i.json 2> $O/mt_$i.err
    timeout 600 python bench.py --scene mesh1m --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 4 > $O/mesh_$i.json 2> $O/mesh_$i.err
    python -c "
import json
a=json.load(open('$O/inst_$i.json')); m=json.load(open('$O/mt_$i.json')); b=json.load(open('$O/mesh_$i.json'))
print('%-5s'%'$v', 'instances10k', a['value'], {k: round(x['avg_us']) for k, x in a['kernels'].items()}, '| materialtest', m['value'], '| mesh1m', b['value'])" 2>&1 | tail -1
    i=$((i+1))
  done
done
