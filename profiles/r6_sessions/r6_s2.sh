#!/bin/bash
# round 6, session 2: baseline bench + counters of all four scenes (session 1 lost the device after a rocprofv3 pass over the micro-benchmark)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s2; mkdir -p $O
alive() { timeout 300 python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)"; }
alive || { echo "no device at start" > $O/DEAD; exit 1; }
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
alive || { echo "no device after bench" > $O/DEAD; exit 1; }
for sc in materialtest cornell instances10k mesh1m; do
  timeout 1500 python tools/pmc_variants.py --out $O/sq_counters_$sc.json --scene $sc --spp 32 --groups lane,sq,mem,tcp,tcc,ifetch --timeout 200 > $O/sq_counters_$sc.txt 2>&1
  alive || { echo "no device after pmc $sc" > $O/DEAD; exit 1; }
done
timeout 400 python tools/pmc_clock.py --out $O/clock.json --scene materialtest --spp 32 > /dev/null 2> $O/clock.err
for sc in cornell instances10k mesh1m; do
  timeout 600 python bench.py --scene $sc --no-cpu-baseline > $O/bench_$sc.json 2> $O/bench_$sc.err
done
TG_SCALE_RESIDUAL_WRITE=$PWD/$O/scale_residual.json TG_SCALE_TABLE=$PWD/$O/device_scale.jsonl timeout 900 python -m pytest tests/test_gpu_scale.py -q -k "above_golden_size" > $O/scale_write.txt 2>&1
TG_SCALE_OPTS="decouple=0,hoist_quad=0" TG_SCALE_RESIDUAL_WRITE=$PWD/$O/scale_residual_sequential.json timeout 900 python -m pytest tests/test_gpu_scale.py -q -k "above_golden_size and (mesh1m or materialtest_sobol or cornell_bump)" > $O/scale_write_seq.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "reduce or rank_comm or host_threads" > $O/tests_reduce.txt 2>&1
timeout 120 tools/bin/ubench_lanes masks > $O/ubench_lane_masks.txt 2>&1
ls -la $O
