#!/bin/bash
out=gpurun_out/disk2; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $out/pytest.log
for i in 1 2; do
timeout 300 python bench.py --scene cornell --spp 256 --no-extra --no-cpu-baseline | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cornell',d['value'],d['ms_per_step'],d['result_ok'])"
done
timeout 300 python bench.py --scene materialtest --spp 64 --no-extra --no-cpu-baseline | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('materialtest',d['value'],d['ms_per_step'],d['result_ok'],{k:v['avg_us'] for k,v in d['kernels'].items()})"
timeout 300 python bench.py --scene mesh1m --spp 32 --no-extra --no-cpu-baseline | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('mesh1m',d['value'],d['ms_per_step'],d['result_ok'],{k:v['avg_us'] for k,v in d['kernels'].items()})"
