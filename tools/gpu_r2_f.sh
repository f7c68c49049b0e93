#!/bin/bash
# round 2, session f: the whole GPU suite + smoke + the default bench line
out=gpurun_out/r2f
mkdir -p $out
export TMPDIR=/tmp
rm -f $out/diverge.jsonl
TG_DIVERGE_TABLE=$PWD/$out/diverge.jsonl timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-1200 $out/bench_default.json; tail -3 $out/bench_default.err
