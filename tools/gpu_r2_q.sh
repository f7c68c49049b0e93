#!/bin/bash
# round 2, session q: workgroups per CU with the two half-pools co-resident (thread counts pinned)
out=gpurun_out/r2q
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
)"; tail -2 $out/$name.err | cut -c1-300; }
PIN="--opt threads_closest=192 --opt threads_shadow=256 --opt threads_shade_simple=192 --opt threads_shade_complex=128"
run mt_bpc4 $B
for b in 5 6 8; do run mt_bpc$b $B --opt blocks_per_cu=$b $PIN; done
run mt_bpc6_auto $B --opt blocks_per_cu=6
run mt_bpc8_auto $B --opt blocks_per_cu=8
run mt_bpc6_s1 $B --opt blocks_per_cu=6 --opt streams=1 $PIN
run m1_bpc4 $B --scene mesh1m --spp 32
run m1_bpc6 $B --scene mesh1m --spp 32 --opt blocks_per_cu=6 $PIN
