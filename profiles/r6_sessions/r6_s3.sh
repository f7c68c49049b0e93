#!/bin/bash
# round 6, session 3: baseline bench + counters (sessions 1-2 ran bench.py with two HIP runtimes in the process: import order)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s3; mkdir -p $O
timeout 120 tools/bin/ubench_lanes masks > $O/ubench_lane_masks.txt 2>&1
timeout 120 tools/bin/ubench_valu > $O/ubench_valu.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; echo BENCH FAILED; exit 1; }
for sc in materialtest cornell instances10k mesh1m; do
  timeout 700 python tools/pmc_variants.py --out $O/sq_counters_$sc.json --scene $sc --spp 32 --groups lane,sq,tcp,tcc --timeout 150 > $O/sq_counters_$sc.txt 2>&1
done
timeout 300 python tools/pmc_clock.py --out $O/clock.json --scene materialtest --spp 32 > /dev/null 2> $O/clock.err
for sc in cornell instances10k mesh1m; do
  timeout 400 python bench.py --scene $sc --no-cpu-baseline > $O/bench_$sc.json 2> $O/bench_$sc.err
done
TG_SCALE_RESIDUAL_WRITE=$PWD/$O/scale_residual.json TG_SCALE_TABLE=$PWD/$O/device_scale.jsonl timeout 600 python -m pytest tests/test_gpu_scale.py -q -k "above_golden_size" > $O/scale_write.txt 2>&1
TG_SCALE_OPTS="decouple=0,hoist_quad=0" TG_SCALE_RESIDUAL_WRITE=$PWD/$O/scale_residual_sequential.json timeout 400 python -m pytest tests/test_gpu_scale.py -q -k "above_golden_size and (mesh1m or materialtest_sobol or cornell_bump)" > $O/scale_write_seq.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "reduce or rank_comm or host_threads or hinted" > $O/tests_reduce.txt 2>&1
ls -la $O
