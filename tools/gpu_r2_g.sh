#!/bin/bash
# round 2, session g: two-level wide BVH (instances)
out=gpurun_out/r2g
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -x -k "instances or trace_rays" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 2 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| nodes/ray', d['nodes_per_ray'], 'prims/ray', d['prims_per_ray'], 'ok', d['result_ok'], d['image_mean'])
except Exception as e:
    print('ERR', e)
PY
)"; tail -2 $out/$name.err; }
run inst_bvh2 $B --scene instances10k --spp 32 --opt wide_bvh=0
run inst_wide $B --scene instances10k --spp 32
run inst_wide_4k $B --scene instances10k --spp 16 --res 3840x2160
