#!/bin/bash
# round 6, session 50: max-ilp on the remaining shading units (shade_inst*.hip, shade_plastic.hip, shade_full.hip) and on tail.hip: instances10k, as shipped, the dielectric hero (class 2)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s50; mkdir -p $O
i=0
for round in 1 2; do
  for v in prod restilp; do
    if [ $v = prod ]; then unset TUNGSTEN_AMD_LIB; else export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so; fi
    timeout 600 python bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 3 > $O/inst_$i.json 2> $O/inst_$i.err
    timeout 600 python tools/bench_as_shipped.py --repeats 4 > $O/shipped_$i.json 2> $O/shipped_$i.err
    timeout 600 python bench.py --material dielectric --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 6 > $O/diel_$i.json 2> $O/diel_$i.err
    python -c "
import json
a=json.load(open('$O/inst_$i.json')); s=json.loads(open('$O/shipped_$i.json').read().strip().splitlines()[-1]); d=json.load(open('$O/diel_$i.json'))
print('%-8s'%'$v', 'instances10k', a['value'], a['image_mean'][0], '| as shipped', s['msamples_per_s'], '| dielectric', d['value'], d['image_mean'][0])" 2>&1 | tail -1
    i=$((i+1))
  done
done
