#!/bin/bash
# round 2, session r: 8 workgroups per CU (4 per half-pool kernel): workgroup sizes, pool size
out=gpurun_out/r2r
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
)"; }
t() { echo "--opt blocks_per_cu=$1 --opt threads_closest=$2 --opt threads_shadow=$3 --opt threads_shade_simple=$4 --opt threads_shade_complex=$5"; }
run mt_base $B
run mt_A $B $(t 8 192 256 128 128)
run mt_B $B $(t 8 192 192 128 128)
run mt_C $B $(t 8 256 256 128 128)
run mt_D $B $(t 8 192 256 128 64)
run mt_E $B $(t 8 192 256 64 64)
run mt_F $B $(t 8 128 256 128 128)
run mt_G $B $(t 6 192 256 128 128)
run mt_A2M $B $(t 8 192 256 128 128) --opt max_slots=2097152
run mt_A512K $B $(t 8 192 256 128 128) --opt max_slots=524288
run mt_base2 $B
run m1_base $B --scene mesh1m --spp 32
run m1_A $B --scene mesh1m --spp 32 $(t 8 192 256 128 128)
run m1_A2M $B --scene mesh1m --spp 32 $(t 8 192 256 128 128) --opt max_slots=2097152
