#!/bin/bash
# round 6, session 10: as shipped (Sobol + adaptive, 16-spp passes) -- when the host looks at the pool and when it hands over to k_tail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s10; mkdir -p $O
for ci in 0 3 4 5 6; do for tt in 8192 65536 262144 1048576; do
  echo "check_interval=$ci tail_threshold=$tt $(timeout 300 python tools/bench_as_shipped.py --repeats 3 --opt check_interval=$ci --opt tail_threshold=$tt 2>> $O/err.txt)" >> $O/as_shipped_sweep.txt
done; done
cut -c1-330 $O/as_shipped_sweep.txt
TGHIP_VERBOSE=1 timeout 300 python tools/bench_as_shipped.py --repeats 1 > $O/verbose_default.json 2> $O/verbose_default.txt
grep tghip $O/verbose_default.txt | head -20
