#!/bin/bash
# round 6, session 1: lane-utilisation calibration + baseline bench + counters of all four scenes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s1; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 120 tools/bin/ubench_nt_coherence > $O/ubench_nt_coherence.txt 2>&1
tools/bin/ubench_lanes > $O/ubench_lanes.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ul -o ul -- $R/tools/bin/ubench_lanes > /dev/null 2>&1)
python tools/pmc_agg.py --variants $(find /tmp/ul -name '*counter_collection.csv') > $O/ubench_lanes_pmc.json 2>&1
# clocks sampled while the driver-form bench runs
(while true; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" ; sleep 0.5; done) > $O/clocks_during_bench.txt &
SMI=$!
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
kill $SMI
timeout 600 python tools/pmc_clock.py --out $O/clock.json --scene materialtest --spp 32 > /dev/null 2> $O/clock.err
for sc in materialtest cornell mesh1m instances10k; do
  timeout 1200 python tools/pmc_variants.py --out $O/sq_counters_$sc.json --scene $sc --spp 32 --groups lane,sq,mem,tcp,tcc,ifetch,sqc > $O/sq_counters_$sc.txt 2>&1
done
for sc in mesh1m instances10k; do
  timeout 600 python bench.py --scene $sc --no-cpu-baseline > $O/bench_$sc.json 2> $O/bench_$sc.err
done
ls -la $O
