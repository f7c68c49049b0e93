#!/bin/bash
# round 2, session x: pool size
out=gpurun_out/r2x
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'], d.get('wavefront_iterations'))
except Exception as e:
    print('ERR', e)
PY
)"; tail -1 $out/$name.err | cut -c1-200; }
run mt_2M $B
run mt_3M $B --opt max_slots=3145728
run mt_4M $B --opt max_slots=4194304
run mt_1M $B --opt max_slots=1048576
run m1_2M $B --scene mesh1m --spp 32
run m1_4M $B --scene mesh1m --spp 32 --opt max_slots=4194304
run inst_2M $B --scene instances10k --spp 32
run inst_4M $B --scene instances10k --spp 32 --opt max_slots=4194304 
