#!/bin/bash
# round 2, session ar: instanced scenes in two parts at 8 workgroups per CU by default: GPU tests, every bench scene
out=gpurun_out/r2ar
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; TGHIP_VERBOSE=1 timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d.get('wavefront_iterations'))
except Exception as e:
    print('ERR', e)
PY
)"; grep "grid" $out/$name.err | tail -1 | cut -c1-120; }
run inst $B --scene instances10k --spp 32
run inst_s1 $B --scene instances10k --spp 32 --opt streams=1
run inst4k $B --scene instances10k --res 3840x2160 --spp 16
run mt $B
run m1 $B --scene mesh1m --spp 32
