#!/bin/bash
# round 2, session ao: closest-hit rays of instanced scenes on a dynamic-fetch two-level BVH2 kernel
out=gpurun_out/r2ao
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -x -k "inst" > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; TGHIP_VERBOSE=1 timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'], d.get('nodes_per_ray'), d.get('prims_per_ray'))
except Exception as e:
    print('ERR', e)
PY
)"; grep "grid" $out/$name.err | tail -1 | cut -c1-120; }
run inst_dyn $B --scene instances10k --spp 32
run inst_static $B --scene instances10k --spp 32 --opt inst_dyn=0
run inst_dyn128 $B --scene instances10k --spp 32 --opt threads_closest=128
run inst_dyn256 $B --scene instances10k --spp 32 --opt threads_closest=256
run inst_dyn_lb4 $B --scene instances10k --spp 32 --opt leaf_batch=4
run inst_dyn_lb16 $B --scene instances10k --spp 32 --opt leaf_batch=16
