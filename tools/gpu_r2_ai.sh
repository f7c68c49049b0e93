#!/bin/bash
# round 2, session ai: is the GPU saturated?  Two / three independent renders side by side against one.
out=gpurun_out/r2ai
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --no-kernel-timing --steps 8 --warmup 2"
get() { python -c "import json,sys;d=json.loads(open('$1').read());print(d['value'], d['ms_per_step'])"; }
timeout 200 $B > $out/one.json 2>/dev/null; echo "one: $(get $out/one.json)"
for i in 1 2; do (timeout 300 $B > $out/two_$i.json 2>/dev/null) & done; wait
echo "two side by side: $(get $out/two_1.json) | $(get $out/two_2.json)"
for i in 1 2 3; do (timeout 300 $B > $out/three_$i.json 2>/dev/null) & done; wait
echo "three side by side: $(get $out/three_1.json) | $(get $out/three_2.json) | $(get $out/three_3.json)"
for i in 1 2; do (timeout 300 $B --opt max_slots=4194304 > $out/two4_$i.json 2>/dev/null) & done; wait
echo "two side by side, 4 M slots each: $(get $out/two4_1.json) | $(get $out/two4_2.json)"
