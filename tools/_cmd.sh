#!/bin/bash
for t in 256 192 384; do
timeout 300 python bench.py --scene materialtest --no-extra --no-cpu-baseline --spp 64 --opt threads_shadow=$t | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('materialtest shadow threads $t',d['value'],d['ms_per_step'],d['result_ok'],{k:v['avg_us'] for k,v in d['kernels'].items()})"
done
timeout 300 python bench.py --scene mesh1m --no-extra --no-cpu-baseline --spp 32 --opt threads_shadow=256 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('mesh1m 256',d['value'],d['ms_per_step'],d['result_ok'],{k:v['avg_us'] for k,v in d['kernels'].items()})"
