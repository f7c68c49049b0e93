#!/bin/bash
# round 2, session aa: 4096 slots per workgroup / 8 M slots by default: GPU tests, every bench scene
out=gpurun_out/r2aa
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
)"; tail -1 $out/$name.err | grep -v amdgpu.ids | cut -c1-200; }
run mt $B
run m1 $B --scene mesh1m --spp 32
run inst $B --scene instances10k --spp 32
run inst_2k $B --scene instances10k --spp 32 --opt slots_per_block=2048
run cornell $B --scene cornell
run mt_64 $B --spp 64
timeout 300 python tools/bench_as_shipped.py > $out/as_shipped.log 2>&1; tail -1 $out/as_shipped.log
