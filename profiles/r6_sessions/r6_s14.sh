#!/bin/bash
# round 6, session 14: the oracle walks masters through their wide subtrees in renders (as the device does) -- the instanced parity tests again; rocprofv3 kernel stats as CSV
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s14; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -k "instance or instances or instanced or c5" > $O/gpu_instances.txt 2>&1
grep -n "passed\|failed" $O/gpu_instances.txt | tail -2
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_mt -o mt -- python $R/bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --steps 2 > $R/$O/prof_mt.json 2> $R/$O/prof_mt.err )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_inst -o inst -- python $R/bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --no-exclusive --steps 2 > $R/$O/prof_inst.json 2> $R/$O/prof_inst.err )
find $O -name "*kernel_stats.csv" | while read f; do cp "$f" $O/$(basename $(dirname $(dirname "$f")))_$(basename "$f") 2>/dev/null || cp "$f" $O/; done
find $O/prof_mt $O/prof_inst -type f -delete 2>/dev/null
ls -la $O; head -12 $O/*kernel_stats.csv | cut -c1-200
