#!/bin/bash
# round 2, session av: short batches on one stream: scheduling / adaptive tests, as-shipped bench, one rank's share of an 8-GPU run
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adaptive.py -m gpu -q --timeout 150 -x -k "scheduling or adaptive or passes or shards or integrator" 2>&1 | grep -E "passed|failed|Error" | tail -2
echo "as_shipped: $(timeout 60 python tools/bench_as_shipped.py --repeats 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['msamples_per_s'], d['seconds'], d['count_min'], d['count_max'], d['image_mean'])")"
timeout 60 python bench.py --no-extra --no-cpu-baseline --no-kernel-timing --no-traffic --steps 3 --warmup 1 --emulate-shards 8 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('shard 1/8', d['ms_per_step'], d['result_ok'])"
