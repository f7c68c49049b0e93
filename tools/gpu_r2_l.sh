#!/bin/bash
# round 2, session l: workgroup-size sweeps of the wide kernels and the shading launches
out=gpurun_out/r2l
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'])
except Exception as e:
    print('ERR', e)
PY
)"; }
run mt_base $B
run mt_c128 $B --opt threads_closest=128
run mt_c192 $B --opt threads_closest=192
run mt_c256 $B --opt threads_closest=256
run mt_s128 $B --opt threads_shadow=128
run mt_s192 $B --opt threads_shadow=192
run mt_ss128 $B --opt threads_shade_simple=128
run mt_ss256 $B --opt threads_shade_simple=256
run mt_sc64 $B --opt threads_shade_complex=64
run mt_base_b $B
run mt_combo $B --opt threads_closest=256 --opt threads_shadow=192
run m1_base $B --scene mesh1m --spp 32
run m1_c256 $B --scene mesh1m --spp 32 --opt threads_closest=256
run m1_c192 $B --scene mesh1m --spp 32 --opt threads_closest=192
run m1_s192 $B --scene mesh1m --spp 32 --opt threads_shadow=192
