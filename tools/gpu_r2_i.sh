#!/bin/bash
# round 2, session i: slot-record pool layout against the array layout, same box, alternating
out=gpurun_out/r2i
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
for rep in 1 2; do
run mt_soa$rep $B
run mt_rec$rep $B --opt pool_layout=1
done
run m1_soa $B --scene mesh1m --spp 32
run m1_rec $B --scene mesh1m --spp 32 --opt pool_layout=1
run inst_soa $B --scene instances10k --spp 32
run inst_rec $B --scene instances10k --spp 32 --opt pool_layout=1
run cornell_soa $B --scene cornell
run cornell_rec $B --scene cornell --opt pool_layout=1
