#!/bin/bash
# round 6, session 45: the number of parts of the wavefront loop with more hardware queues than HIP's default four (GPU_MAX_HW_QUEUES), metric's workload
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s45; mkdir -p $O
i=0
for round in 1 2; do
  for cfg in "4 4" "8 4" "8 5" "8 6" "8 8" "2 4" "16 8"; do
    set -- $cfg
    GPU_MAX_HW_QUEUES=$1 timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 10 --opt streams=$2 > $O/b_$i.json 2> $O/b_$i.err
    python -c "
import json; d=json.load(open('$O/b_$i.json')); print('queues $1 parts $2', d['value'], {k: round(v['avg_us']) for k, v in d['kernels'].items()})" 2>&1 | tail -1
    i=$((i+1))
  done
done
