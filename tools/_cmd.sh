#!/bin/bash
out=gpurun_out/ad2; mkdir -p $out; export TMPDIR=/tmp
for i in 1 2; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$i -o stats -- python bench.py --scene materialtest --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing > $out/prof$i.log 2>&1
f=$(find $out/prof$i -name '*kernel_stats.csv' | head -1); head -6 $f | cut -c1-150
find $out/prof$i -name '*kernel_trace.csv' -delete; find $out/prof$i -name '*.db' -delete
done
for i in 1 2 3; do
timeout 300 python bench.py --scene materialtest --spp 64 --no-extra --no-cpu-baseline | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('materialtest',d['value'],d['ms_per_step'],{k:v['avg_us'] for k,v in d['kernels'].items()})"
done
