#!/bin/bash
# round 2, session c: phase voting in the wide kernels
out=gpurun_out/r2c
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| nodes/ray', d['nodes_per_ray'], 'prims/ray', d['prims_per_ray'], 'ok', d['result_ok'], d['image_mean'])
except Exception as e:
    print('ERR', e)
PY
)"; }
run mt_lb1 $B
run mt_lb2 $B --opt leaf_batch=2
run mt_lb3 $B --opt leaf_batch=3
run mt_lb4 $B --opt leaf_batch=4
run mt_lb8 $B --opt leaf_batch=8
run m1_lb1 $B --scene mesh1m --spp 32
run m1_lb2 $B --scene mesh1m --spp 32 --opt leaf_batch=2
run m1_lb4 $B --scene mesh1m --spp 32 --opt leaf_batch=4
