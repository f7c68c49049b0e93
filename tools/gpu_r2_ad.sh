#!/bin/bash
# round 2, session ad: parts of the pool at 8 M slots
out=gpurun_out/r2ad
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d.get('wavefront_iterations'))
except Exception as e:
    print('ERR', e)
PY
)"; }
run mt_4 $B
run mt_8 $B --opt streams=8
run mt_2 $B --opt streams=2
run mt_8_c128 $B --opt streams=8 --opt threads_closest=128 --opt threads_shadow=128
run m1_4 $B --scene mesh1m --spp 32
run m1_8 $B --scene mesh1m --spp 32 --opt streams=8
run mt64_4 $B --spp 64
run mt64_8 $B --spp 64 --opt streams=8
