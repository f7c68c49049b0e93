#!/bin/bash
# round 2, session au: the as-shipped configuration (short adaptive passes) against parts / check interval / pool
export TMPDIR=/tmp
for o in "" "--opt streams=1" "--opt streams=2" "--opt check_interval=2" "--opt check_interval=8" "--opt streams=2 --opt check_interval=2" "--opt max_slots=2097152 --opt streams=2" "--opt streams=1 --opt blocks_per_cu=8"; do
  echo "as_shipped $o: $(timeout 60 python tools/bench_as_shipped.py --repeats 2 $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['msamples_per_s'], d['seconds'])")"
done
