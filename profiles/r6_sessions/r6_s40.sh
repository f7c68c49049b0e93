#!/bin/bash
# round 6, session 40: the equirectangular camera through k_camera_rays (no kernel that holds nextPath changed) -- goldens per sample (uniform and Sobol'), the reference program
# with the plugin, the scheduling / adaptive tests it could touch; the headline once
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_samples.py tests/test_ref_binding.py -m gpu -q -k "equirect or thinlens or cornell_sobol or cornell_box_filter or cornell_samples" > $O/gpu_equirect.txt 2>&1
tail -6 $O/gpu_equirect.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 12 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r6_s40/bench.json')); print(d['value'], d['kernels']['k_trace_closest']['avg_us'])"
