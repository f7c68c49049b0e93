#!/bin/bash
out=gpurun_out/r2d
mkdir -p $out
export TMPDIR=/tmp
TGHIP_VERBOSE=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 2 --warmup 1 > $out/mt.json 2> $out/mt.err; echo rc=$?
grep -E "wave turns|grid" $out/mt.err | tail -5
python - <<PY
import json
d=json.loads(open('$out/mt.json').read())
print(d['value'], d['kernels'], 'nodes/ray', d['nodes_per_ray'], 'prims', d['prims_per_ray'], 'rays/sample', d['rays_per_sample'], 'iters', d['wavefront_iterations'])
PY
