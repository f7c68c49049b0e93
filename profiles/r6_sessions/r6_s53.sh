#!/bin/bash
# round 6, session 53: instances10k under the other scheduler strategies on the traversal units (max-memory-clause, iterative-minreg, iterative-maxocc); twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s53; mkdir -p $O
i=0
for round in 1 2; do
  for v in prod wmem wiminreg wimaxocc; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 3 > $O/inst_$i.json 2> $O/inst_$i.err
    python -c "
import json
a=json.load(open('$O/inst_$i.json'))
print('%-9s'%'$v', 'instances10k', a['value'], {k: round(x['avg_us']) for k, x in a['kernels'].items()}, a['image_mean'][0])" 2>&1 | tail -1
    i=$((i+1))
  done
done
