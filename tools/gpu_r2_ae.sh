#!/bin/bash
# round 2, session ae: one rank's share of an N-GPU run (--emulate-shards N) against the pool size
out=gpurun_out/r2ae
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
run() { name=$1; shift; timeout 120 "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read())
    print(d['value'], 'Ms/s', d['ms_per_step'], 'ms | ok', d['result_ok'], d.get('wavefront_iterations'))
except Exception as e:
    print('ERR', e)
PY
)"; }
run n1 $B
for n in 2 4 8; do
  run n${n}_8M $B --emulate-shards $n
  run n${n}_4M $B --emulate-shards $n --opt max_slots=4194304
  run n${n}_2M $B --emulate-shards $n --opt max_slots=2097152
done
run n8_1M $B --emulate-shards 8 --opt max_slots=1048576
run n8_2M_c16 $B --emulate-shards 8 --opt max_slots=2097152 --opt chunk_samples=16
run n8_8M_c16 $B --emulate-shards 8 --opt chunk_samples=16
