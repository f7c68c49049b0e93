#!/bin/bash
# round 2, session k: over-subscribed grids (grid_rounds) and workgroup sizes, interleaved repeats on one box
out=gpurun_out/r2k
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
for rep in 1 2; do
run mt_base$rep $B
run mt_r2_$rep $B --opt grid_rounds=2
run mt_r4_$rep $B --opt grid_rounds=4
run mt_c256_$rep $B --opt threads_closest=256
run mt_c256r2_$rep $B --opt threads_closest=256 --opt grid_rounds=2
done
run m1_base $B --scene mesh1m --spp 32
run m1_r2 $B --scene mesh1m --spp 32 --opt grid_rounds=2
run m1_r4 $B --scene mesh1m --spp 32 --opt grid_rounds=4
run inst_base $B --scene instances10k --spp 32
run inst_r2 $B --scene instances10k --spp 32 --opt grid_rounds=2
