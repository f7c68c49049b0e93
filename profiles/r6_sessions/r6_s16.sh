#!/bin/bash
# round 6, session 16: instanced scenes' shadow rays on k_trace_shadow_fast_inst -- parity, A/B; the Cornell box's fused kernel, lanes per section at 64 / 256 spp; smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s16; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -k "instance or instances or instanced or c5" > $O/gpu_instances.txt 2>&1
grep -n "passed\|failed" $O/gpu_instances.txt | tail -2
timeout 900 python tools/sweep.py --scene instances10k --steps 2 --repeat 2 -- "inst_shadow_fast=0" "inst_shadow_fast=1" > $O/sweep_ab.jsonl 2> $O/sweep_ab.err
cut -c1-260 $O/sweep_ab.jsonl
timeout 900 python tools/sweep.py --scene instances10k --steps 2 -- "inst_shadow_fast=1,leaf_batch=1" "inst_shadow_fast=1,leaf_batch=2" "inst_shadow_fast=1,leaf_batch=3" "inst_shadow_fast=1,leaf_batch=4" > $O/sweep_vote.jsonl 2> $O/sweep_vote.err
cut -c1-260 $O/sweep_vote.jsonl
P=$PWD/tungsten_amd/lib/libtungsten_hip_prof.so
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-kernel-timing"
for spp in 64 256; do
  TGHIP_VERBOSE=1 TUNGSTEN_AMD_LIB=$P timeout 300 python bench.py $Q --scene cornell --spp $spp --steps 1 --warmup 0 > $O/prof_cornell_$spp.json 2> $O/prof_cornell_$spp.txt
  grep "class 0" $O/prof_cornell_$spp.txt | tail -1 | cut -c1-700
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
