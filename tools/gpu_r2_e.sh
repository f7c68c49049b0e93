#!/bin/bash
# round 2, session e: trimmed node test (packed FMA, leaf_valid), workgroup-size sweeps of the wide kernels
out=gpurun_out/r2e
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "trace_rays or materialtest or mesh1m or water" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
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
run mt_base $B
run mt_c192 $B --opt threads_closest=192
run mt_c256 $B --opt threads_closest=256
run mt_c384 $B --opt threads_closest=384
run mt_s192 $B --opt threads_shadow=192
run mt_s320 $B --opt threads_shadow=320
run mt_s384 $B --opt threads_shadow=384
run mt_b5 $B --opt blocks_per_cu=5
run mt_b6 $B --opt blocks_per_cu=6
run mt_b8 $B --opt blocks_per_cu=8
run m1_base $B --scene mesh1m --spp 32
