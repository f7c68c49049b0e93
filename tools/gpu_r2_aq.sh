#!/bin/bash
# round 2, session aq: parts / workgroups per CU for the instanced scene with the dynamic-fetch closest-hit kernel
out=gpurun_out/r2aq
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1 --scene instances10k --spp 32"
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
PIN="--opt threads_closest=192 --opt threads_shadow=256 --opt threads_shade_simple=128 --opt threads_shade_complex=128"
run base $B
run s2 $B --opt streams=2
run s2_b8 $B --opt streams=2 --opt blocks_per_cu=8 $PIN
run s4_b8 $B --opt streams=4 --opt blocks_per_cu=8 $PIN
run s1_b8 $B --opt blocks_per_cu=8 $PIN
run s2_b6 $B --opt streams=2 --opt blocks_per_cu=6 $PIN
