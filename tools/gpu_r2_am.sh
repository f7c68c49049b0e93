#!/bin/bash
# round 2, session am: instanced wide walk, one thing per turn, leave / resume without turns of their own
out=gpurun_out/r2am
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'], d.get('nodes_per_ray'), d.get('prims_per_ray'))
except Exception as e:
    print('ERR', e)
PY
)"; }
run inst $B --scene instances10k --spp 32
run inst_wc $B --scene instances10k --spp 32 --opt wide_closest=1
run inst_wc256 $B --scene instances10k --spp 32 --opt wide_closest=1 --opt threads_closest=256
run inst_wc128 $B --scene instances10k --spp 32 --opt wide_closest=1 --opt threads_closest=128
