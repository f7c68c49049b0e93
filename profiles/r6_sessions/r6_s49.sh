#!/bin/bash
# round 6, session 49: the traversal translation units (tungsten_hip.hip, walk_shadow.hip) under -O2 and the AMDGPU scheduler strategies iterative-minreg / iterative-maxocc against the product's,
# metric's workload and mesh1m; the list twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s49; mkdir -p $O
i=0
for round in 1 2; do
  for v in prod wo2 wiminreg wimaxocc; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 10 > $O/mt_$i.json 2> $O/mt_$i.err
    timeout 600 python bench.py --scene mesh1m --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 4 > $O/mesh_$i.json 2> $O/mesh_$i.err
    python -c "
import json
a=json.load(open('$O/mt_$i.json')); b=json.load(open('$O/mesh_$i.json'))
print('%-9s'%'$v', 'materialtest', a['value'], {k: round(x['avg_us']) for k, x in a['kernels'].items()}, a['image_mean'][0], '| mesh1m', b['value'], b['image_mean'][0])" 2>&1 | tail -1
    i=$((i+1))
  done
done
