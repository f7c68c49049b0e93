#!/bin/bash
# counter passes over the materialtest kernels (wide BVH): where do the waves spend their time, how do the caches do
out=gpurun_out/r2pmc
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$out/counters.txt 2>&1
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --no-kernel-timing --steps 1 --warmup 0 --spp 32"
pass() { name=$1; ctrs=$2; shift 2; timeout 400 rocprofv3 --pmc $ctrs --output-format csv -d $out/$name -o pmc -- $B "$@" > $out/$name.log 2>&1; echo "$name rc=$?"
  f=$(find $out/$name -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_agg.py $f > $out/$name.json; rm -rf $out/$name; }
pass sqA "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
pass sqB "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
pass tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
pass tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
pass sqA_bvh2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" --opt wide_bvh=0
pass tcc_bvh2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" --opt wide_bvh=0
ls -la $out
