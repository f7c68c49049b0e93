#!/bin/bash
# round 2, session n: record + next node fetched in one turn (wide kernels) against one thing per turn (leaf_batch=9), alternating
out=gpurun_out/r2n
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py -m gpu -q --timeout 600 -x -k "materialtest or mesh1m or water or zoo_a or trace_rays" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| nodes/ray', d['nodes_per_ray'], 'prims/ray', d['prims_per_ray'], 'ok', d['result_ok'], d['image_mean'])
except Exception as e:
    print('ERR', e)
PY
)"; }
for rep in 1 2; do
run mt_single$rep $B --opt leaf_batch=9
run mt_dual$rep $B
done
run m1_single $B --scene mesh1m --spp 32 --opt leaf_batch=9
run m1_dual $B --scene mesh1m --spp 32
run mt_dual_two $B --opt streams=2
