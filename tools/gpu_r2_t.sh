#!/bin/bash
# round 2, session t: the new defaults (8 workgroups per CU, two parts) under the GPU tests; 3 and 4 parts
out=gpurun_out/r2t
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
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
run mt_2 $B
run mt_3 $B --opt streams=3
run mt_4 $B --opt streams=4
run mt_4b $B --opt streams=4 --opt threads_closest=192
run mt_2b $B
run m1_2 $B --scene mesh1m --spp 32
run m1_4 $B --scene mesh1m --spp 32 --opt streams=4
run inst $B --scene instances10k --spp 32
