#!/bin/bash
# round 2, session a: node-walk microbenchmark, the GPU suite (with the per-sample divergence table), the default bench line
out=gpurun_out/r2a
mkdir -p $out
export TMPDIR=/tmp
echo "== ubench"
timeout 300 tools/bin/ubench_chase > $out/ubench.txt 2>&1; echo "rc=$?"; tail -5 $out/ubench.txt
echo "== pytest -m gpu"
rm -f $out/diverge.jsonl
TG_DIVERGE_TABLE=$PWD/$out/diverge.jsonl timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $out/pytest.log
echo "== bench default"
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc=$?"; cut -c1-1500 $out/bench_default.json; tail -3 $out/bench_default.err
