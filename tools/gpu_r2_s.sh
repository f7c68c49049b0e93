#!/bin/bash
# round 2, session s: around 8 workgroups per CU / closest 256 / shadow 256 / shade 128
out=gpurun_out/r2s
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
run mt_C $B $(t 8 256 256 128 128)
run mt_C1 $B $(t 8 256 256 192 128)
run mt_C2 $B $(t 8 256 320 128 128)
run mt_C3 $B $(t 8 320 256 128 128)
run mt_C4 $B $(t 8 256 256 128 192)
run mt_C5 $B $(t 8 256 256 256 128)
run mt_C6 $B $(t 7 256 256 128 128)
run m1_C $B --scene mesh1m --spp 32 $(t 8 256 256 128 128)
run m1_C1 $B --scene mesh1m --spp 32 $(t 8 256 256 192 128)
run inst_base $B --scene instances10k --spp 32
run inst_C $B --scene instances10k --spp 32 $(t 8 256 256 128 128)
run inst_C_s2 $B --scene instances10k --spp 32 $(t 8 256 256 128 128) --opt streams=2
run inst_s2 $B --scene instances10k --spp 32 --opt streams=2
run cornell $B --scene cornell
