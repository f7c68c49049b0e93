#!/bin/bash
# round 2, session o: regeneration left to k_finish (vs in k_shade), alone and with the paired fetch + two streams
out=gpurun_out/r2o
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py tests/test_gpu_adaptive.py tests/test_gpu_outputs.py tests/test_media.py -m gpu -q --timeout 600 -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
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
for rep in 1 2; do
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_base/libtungsten_hip.so run mt_base$rep $B
run mt_defer$rep $B --opt streams=1 --opt leaf_batch=9
run mt_all$rep $B
done
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_base/libtungsten_hip.so run m1_base $B --scene mesh1m --spp 32
run m1_defer $B --scene mesh1m --spp 32 --opt streams=1 --opt leaf_batch=9
run m1_all $B --scene mesh1m --spp 32
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_base/libtungsten_hip.so run inst_base $B --scene instances10k --spp 32
run inst_all $B --scene instances10k --spp 32
