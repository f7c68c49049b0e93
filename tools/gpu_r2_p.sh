#!/bin/bash
# round 2, session p: how evenly is k_shade's work spread (PT_PROFILE build, per class)?  work-item groups of 16; tight instance boxes
out=gpurun_out/r2p
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    k=d['kernels']
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms |', ' '.join('%s %.0fus'%(n.replace('k_trace_',''),k[n]['avg_us']) for n in k), '| ok', d['result_ok'], d['image_mean'], 'nodes/ray', d.get('nodes_per_ray'))
except Exception as e:
    print('ERR', e)
PY
)"; }
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_prof/libtungsten_hip.so run mt_prof $B --spp 64 --opt streams=1
grep PT_PROFILE $out/mt_prof.err | tail -4
run mt_main1 $B
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_g16/libtungsten_hip.so run mt_g16 $B
run mt_main2 $B
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_g16/libtungsten_hip.so run m1_g16 $B --scene mesh1m --spp 32
run m1_main $B --scene mesh1m --spp 32
run inst_tight $B --scene instances10k --spp 32
TGH_LOOSE_INSTANCE_BOUNDS=1 run inst_loose $B --scene instances10k --spp 32
