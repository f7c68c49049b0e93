#!/bin/bash
# round 2, session b: the 8-wide BVH kernels -- parity first, then A/B timings against the BVH2 kernels
out=gpurun_out/r2b
mkdir -p $out
export TMPDIR=/tmp
echo "== pytest (trace_rays exact counts, parity, per-sample)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py -m gpu -q --timeout 600 -x -k "not non_exponential" > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| nodes/ray', d['nodes_per_ray'], 'prims/ray', d['prims_per_ray'])
except Exception as e:
    print('ERR', e)
PY
)"; }
run mt_bvh2 $B --opt wide_bvh=0
run mt_wide80 $B
run mt_wide128 $B --opt wide_node_stride=128
TGH_WIDE_PRIM_COST=0.3 run mt_wide80_pc03 $B
TGH_WIDE_PRIM_COST=1.0 run mt_wide80_pc10 $B
TGH_WIDE_PRIM_COST=1.0 run mt_wide128_pc10 $B --opt wide_node_stride=128
run m1_bvh2 $B --scene mesh1m --spp 32 --opt wide_bvh=0
run m1_wide80 $B --scene mesh1m --spp 32
run m1_wide128 $B --scene mesh1m --spp 32 --opt wide_node_stride=128
