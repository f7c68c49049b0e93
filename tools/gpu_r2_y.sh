#!/bin/bash
# round 2, session y: larger pools (slots per workgroup 4096 / 8192 builds; grid_rounds)
out=gpurun_out/r2y
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'], d.get('wavefront_iterations'))
except Exception as e:
    print('ERR', e)
PY
)"; tail -1 $out/$name.err | grep -v amdgpu.ids | cut -c1-200; }
S4=$PWD/tungsten_amd/lib_s4k/libtungsten_hip.so
S8=$PWD/tungsten_amd/lib_s8k/libtungsten_hip.so
run mt_4M $B --opt max_slots=4194304
run mt_4M_r2 $B --opt max_slots=8388608 --opt grid_rounds=2
TUNGSTEN_AMD_LIB=$S4 run mt_s4k_4M $B --opt max_slots=4194304
TUNGSTEN_AMD_LIB=$S4 run mt_s4k_8M $B --opt max_slots=8388608
TUNGSTEN_AMD_LIB=$S8 run mt_s8k_8M $B --opt max_slots=8388608
TUNGSTEN_AMD_LIB=$S8 run mt_s8k_16M $B --opt max_slots=16777216
TUNGSTEN_AMD_LIB=$S4 run m1_s4k_8M $B --scene mesh1m --spp 32 --opt max_slots=8388608
TUNGSTEN_AMD_LIB=$S8 run m1_s8k_16M $B --scene mesh1m --spp 32 --opt max_slots=16777216
TUNGSTEN_AMD_LIB=$S4 run inst_s4k_4M $B --scene instances10k --spp 32 --opt max_slots=4194304
TUNGSTEN_AMD_LIB=$S8 run inst_s8k_8M $B --scene instances10k --spp 32 --opt max_slots=8388608
