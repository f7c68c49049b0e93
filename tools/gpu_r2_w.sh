#!/bin/bash
# round 2, session w: upper bound of keeping the envmap tables cache-resident (throwaway builds that read a small window of them)
out=gpurun_out/r2w
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'], d.get('rays_per_sample'))
except Exception as e:
    print('ERR', e)
PY
)"; }
run mt_main $B
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_exp1/libtungsten_hip.so run mt_texwin $B
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_exp3/libtungsten_hip.so run mt_allwin $B
run mt_main_s1 $B --opt streams=1
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_exp1/libtungsten_hip.so run mt_texwin_s1 $B --opt streams=1
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_exp3/libtungsten_hip.so run mt_allwin_s1 $B --opt streams=1
