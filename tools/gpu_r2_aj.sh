#!/bin/bash
# round 2, session aj: wide kernels squeezed to 5 / 6 waves per SIMD; the Cornell line's VALU roofline
out=gpurun_out/r2aj
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d.get('wavefront_iterations'))
except Exception as e:
    print('ERR', e)
PY
)"; }
for v in main w5c w5cs w6c main; do
  L=$PWD/tungsten_amd/lib_$v/libtungsten_hip.so; [ $v = main ] && L=$PWD/tungsten_amd/lib/libtungsten_hip.so
  TUNGSTEN_AMD_LIB=$L run mt_$v $B
done
for v in main w5c w5cs; do
  L=$PWD/tungsten_amd/lib_$v/libtungsten_hip.so; [ $v = main ] && L=$PWD/tungsten_amd/lib/libtungsten_hip.so
  TUNGSTEN_AMD_LIB=$L run m1_$v $B --scene mesh1m --spp 32
done
timeout 300 python bench.py --scene cornell --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/cornell.json 2> $out/cornell.err; python -c "
import json;d=json.loads(open('$out/cornell.json').read());print('cornell', d['value'], d['roofline'].get('valu'), d['roofline'].get('traffic'))"
