#!/bin/bash
# round 2, session h: envmap-sampling chain (LDS marginals + interleaved row pairs) against the previous build, same box
out=gpurun_out/r2h
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py -m gpu -q --timeout 600 -x -k "materialtest or mesh1m or instances or zoo_a or sun_sky" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
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
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_prev.so run mt_prev$rep $B
run mt_new$rep $B
done
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_prev.so run m1_prev $B --scene mesh1m --spp 32
run m1_new $B --scene mesh1m --spp 32
