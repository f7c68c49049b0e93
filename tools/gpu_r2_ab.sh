#!/bin/bash
# round 2, session ab: the as-shipped (adaptive, Sobol, 16-spp passes) configuration and the instanced scene against pool size
out=gpurun_out/r2ab
mkdir -p $out
export TMPDIR=/tmp
for o in "" "--opt max_slots=2097152" "--opt max_slots=4194304" "--opt max_slots=2097152 --opt slots_per_block=2048" "--opt chunk_samples=2" "--opt chunk_samples=8"; do
  echo "as_shipped $o: $(timeout 300 python tools/bench_as_shipped.py $o 2>&1 | tail -1 | cut -c100-330)"
done
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
)"; grep "grid" $out/$name.err | tail -1 | cut -c1-200; }
run inst $B --scene instances10k --spp 32
run inst_2k $B --scene instances10k --spp 32 --opt slots_per_block=2048
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib_g16/libtungsten_hip.so run inst_oldlib $B --scene instances10k --spp 32
run inst_c384 $B --scene instances10k --spp 32 --opt threads_closest=384
run inst_c512 $B --scene instances10k --spp 32 --opt threads_closest=512
run inst_c256 $B --scene instances10k --spp 32 --opt threads_closest=256
