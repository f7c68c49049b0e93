#!/bin/bash
# One GPU-box session: parity tests, bench lines, rocprofv3 kernel stats.  Usage: tools/gpu_session.sh <tag>
tag=${1:-session}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $out/pytest.log
echo "== bench cornell"
timeout 600 python bench.py > $out/bench_cornell.json 2> $out/bench_cornell.err; echo "rc=$?"; cat $out/bench_cornell.json; tail -5 $out/bench_cornell.err
echo "== bench materialtest"
timeout 600 python bench.py --scene materialtest --spp 64 > $out/bench_materialtest.json 2> $out/bench_materialtest.err; echo "rc=$?"; cat $out/bench_materialtest.json; tail -5 $out/bench_materialtest.err
echo "== rocprof cornell"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o cornell -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/prof_cornell.log 2>&1; echo "rc=$?"
cat $out/prof/cornell_kernel_stats.csv 2>/dev/null | head -12
echo "== rocprof materialtest"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o materialtest -- python bench.py --scene materialtest --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/prof_materialtest.log 2>&1; echo "rc=$?"
cat $out/prof/materialtest_kernel_stats.csv 2>/dev/null | head -12
rm -f $out/prof/*results.db $out/prof/*kernel_trace.csv
