#!/bin/bash
# round 2, session an: classes 0 / 2 of instanced scenes on a MASK_SIMPLE | FEAT_INSTANCES variant
out=gpurun_out/r2an
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -x -k "inst" > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $out/pytest.log
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
run inst_simple $B --scene instances10k --spp 32
run inst_full $B --scene instances10k --spp 32 --opt inst_simple=0
run inst_simple192 $B --scene instances10k --spp 32 --opt threads_shade_simple=192
run inst_simple256 $B --scene instances10k --spp 32 --opt threads_shade_simple=256
