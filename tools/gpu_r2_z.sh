#!/bin/bash
# round 2, session z: 8 M slots -- 4096 per workgroup against twice the workgroups
out=gpurun_out/r2z
mkdir -p $out
export TMPDIR=/tmp
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
S4=$PWD/tungsten_amd/lib_s4k/libtungsten_hip.so
run mt_4M $B --opt max_slots=4194304
run mt_8M_r2 $B --opt max_slots=8388608 --opt grid_rounds=2
TUNGSTEN_AMD_LIB=$S4 run mt_s4k_8M $B --opt max_slots=8388608
TUNGSTEN_AMD_LIB=$S4 run mt_s4k_8M_256 $B --opt max_slots=8388608 --opt threads_shade_simple=256 --opt threads_shade_complex=256
run m1_4M $B --scene mesh1m --spp 32 --opt max_slots=4194304
run m1_8M_r2 $B --scene mesh1m --spp 32 --opt max_slots=8388608 --opt grid_rounds=2
TUNGSTEN_AMD_LIB=$S4 run m1_s4k_8M $B --scene mesh1m --spp 32 --opt max_slots=8388608
run inst_2M $B --scene instances10k --spp 32
run inst_4M_r2 $B --scene instances10k --spp 32 --opt max_slots=4194304 --opt grid_rounds=2
run inst_8M_r4 $B --scene instances10k --spp 32 --opt max_slots=8388608 --opt grid_rounds=4
