#!/bin/bash
# round 2, session u: four parts by default, record (adaptive) passes split as well
out=gpurun_out/r2u
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
run mt_4 $B
run mt_2 $B --opt streams=2
timeout 300 python tools/bench_as_shipped.py > $out/as_shipped.log 2>&1; tail -3 $out/as_shipped.log
timeout 300 python tools/bench_as_shipped.py --opt streams=1 > $out/as_shipped_s1.log 2>&1; tail -3 $out/as_shipped_s1.log
