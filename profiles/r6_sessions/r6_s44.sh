#!/bin/bash
# round 6, session 44: HIP stream priorities of the four parts' streams (TGHIP_STREAM_PRIORITIES: main, part 1..3) on the metric's workload, alternated
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s44; mkdir -p $O
python - <<'PY'
import ctypes as C
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(0), C.c_int(0)
print("hipDeviceGetStreamPriorityRange rc", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "least", lo.value, "greatest", hi.value)
PY
i=0
for round in 1 2; do
  for p in none "0,0,0,0" "-1,-1,-1,-1" "-1,0,0,1" "-1,-1,1,1" "1,0,0,-1" "0,-1,0,-1"; do
    if [ "$p" = none ]; then unset TGHIP_STREAM_PRIORITIES; else export TGHIP_STREAM_PRIORITIES="$p"; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 10 > $O/b_$i.json 2> $O/b_$i.err
    python -c "
import json; d=json.load(open('$O/b_$i.json')); print('%-14s'%'$p', d['value'], {k: round(v['avg_us']) for k, v in d['kernels'].items()})"
    i=$((i+1))
  done
done
