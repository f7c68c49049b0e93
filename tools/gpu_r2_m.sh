#!/bin/bash
# round 2, session m: two halves of the pool on two streams against one stream, alternating
out=gpurun_out/r2m
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'])
except Exception as e:
    print('ERR', e)
PY
)"; tail -1 $out/$name.err | cut -c1-200; }
for rep in 1 2; do
run mt_one$rep $B
run mt_two$rep $B --opt streams=2
done
run m1_one $B --scene mesh1m --spp 32
run m1_two $B --scene mesh1m --spp 32 --opt streams=2
run inst_one $B --scene instances10k --spp 32
run inst_two $B --scene instances10k --spp 32 --opt streams=2
