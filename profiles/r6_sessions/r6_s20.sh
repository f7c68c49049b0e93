#!/bin/bash
# round 6, session 20: sections of the one-launch render with PT_TRACE_AHEAD (PT_PROFILE build)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s20; mkdir -p $O
P=$PWD/tungsten_amd/lib/libtungsten_hip_prof.so
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-kernel-timing"
TGHIP_VERBOSE=1 TUNGSTEN_AMD_LIB=$P timeout 300 python bench.py $Q --scene cornell --spp 256 --steps 1 --warmup 0 > $O/prof_cornell_256.json 2> $O/prof_cornell_256.txt
grep "class 0" $O/prof_cornell_256.txt | tail -1 | cut -c1-800
