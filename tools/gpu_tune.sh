#!/bin/bash
# Option sweep on the GPU box (one bench line per configuration).  Usage: tools/gpu_tune.sh <tag> <scene> <spp> "<opts>" ...
tag=$1; scene=$2; spp=$3; shift 3
out=gpurun_out/$tag
mkdir -p $out
for o in "$@"; do
  args=""
  for kv in $o; do args="$args --opt $kv"; done
  echo "== $scene $o"
  timeout 300 python bench.py --scene $scene --spp $spp --res $([ $scene = mesh1m ] && echo 1920x1080 || echo 1280x720) --steps 2 --warmup 1 --no-cpu-baseline --no-extra $args 2>>$out/err.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%8.1f Msamples/s  %7.2f ms/step  iters %d  kernels %s' % (d['value'], d['ms_per_step'], d['wavefront_iterations'], {k: (v['avg_us'], v['gbs']) for k, v in d['kernels'].items()}))
" | tee -a $out/tune_$scene.log
done
