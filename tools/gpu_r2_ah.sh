#!/bin/bash
# round 2, session ah: host check interval, samples per work item at the 8 M pool
out=gpurun_out/r2ah
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
run mt $B
run mt_ci8 $B --opt check_interval=8
run mt_ci16 $B --opt check_interval=16
run mt_ci2 $B --opt check_interval=2
run mt_cs8 $B --opt chunk_samples=8
run mt_cs16 $B --opt chunk_samples=16
run mt_cs2 $B --opt chunk_samples=2
run m1_ci16 $B --scene mesh1m --spp 32 --opt check_interval=16
run m1 $B --scene mesh1m --spp 32
run inst_ci16 $B --scene instances10k --spp 32 --opt check_interval=16
