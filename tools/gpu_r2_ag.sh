#!/bin/bash
# round 2, session ag: hardware queues (GPU_MAX_HW_QUEUES) under the part streams and the class streams
out=gpurun_out/r2ag
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
run mt_c0 $B --opt class_streams=0
run mt_c1 $B
for q in 2 8 16 24; do
GPU_MAX_HW_QUEUES=$q run mt_c0_q$q $B --opt class_streams=0
GPU_MAX_HW_QUEUES=$q run mt_c1_q$q $B
done
GPU_MAX_HW_QUEUES=16 run m1_c1_q16 $B --scene mesh1m --spp 32
GPU_MAX_HW_QUEUES=16 run m1_c0_q16 $B --scene mesh1m --spp 32 --opt class_streams=0
run m1_c0 $B --scene mesh1m --spp 32 --opt class_streams=0
