#!/bin/bash
# round 2, session ap: leaf batch of the dynamic-fetch two-level BVH2 kernel
out=gpurun_out/r2ap
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'])
except Exception as e:
    print('ERR', e)
PY
)"; }
run inst_static $B --scene instances10k --spp 32 --opt inst_dyn=0
for lb in 8 16 24 32 48; do run inst_lb$lb $B --scene instances10k --spp 32 --opt leaf_batch_bvh2=$lb; done
run inst_lb16_t128 $B --scene instances10k --spp 32 --opt threads_closest=128
run inst_lb16_t256 $B --scene instances10k --spp 32 --opt threads_closest=256
run inst_lb32_t256 $B --scene instances10k --spp 32 --opt threads_closest=256 --opt leaf_batch_bvh2=32
run inst_full $B --scene instances10k --spp 32 --opt inst_simple=0
