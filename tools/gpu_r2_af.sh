#!/bin/bash
# round 2, session af: the shading classes of an iteration side by side on three streams
out=gpurun_out/r2af
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'], d.get('wavefront_iterations'))
except Exception as e:
    print('ERR', e)
PY
)"; }
run mt_c1 $B
run mt_c0 $B --opt class_streams=0
run mt_c1b $B
run m1_c1 $B --scene mesh1m --spp 32
run m1_c0 $B --scene mesh1m --spp 32 --opt class_streams=0
run inst_c1 $B --scene instances10k --spp 32
run inst_c0 $B --scene instances10k --spp 32 --opt class_streams=0
run mt64_c1 $B --spp 64
run mt64_c0 $B --spp 64 --opt class_streams=0
