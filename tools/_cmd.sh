#!/bin/bash
out=gpurun_out/direct1; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $out/pytest.log
for i in 1 2; do
timeout 300 python bench.py --scene cornell --spp 256 --no-extra --no-cpu-baseline | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cornell',d['value'],d['ms_per_step'],d['result_ok'],{k:v['avg_us'] for k,v in d['kernels'].items()})"
done
timeout 300 python bench.py --scene cornell --spp 256 --no-extra --no-cpu-baseline --emulate-shards 8 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cornell shard 1/8',d['value'],d['ms_per_step'])"
