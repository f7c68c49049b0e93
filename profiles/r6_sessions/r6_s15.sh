#!/bin/bash
# round 6, session 15: the Cornell box's fused kernel, lanes per section at 64 and at 256 spp (PT_PROFILE build); smoke() and build() as the driver runs them
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s15; mkdir -p $O
P=$PWD/tungsten_amd/lib/libtungsten_hip_prof.so
Q="--no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-kernel-timing"
for spp in 64 256; do
  TGHIP_VERBOSE=1 TUNGSTEN_AMD_LIB=$P timeout 300 python bench.py $Q --scene cornell --spp $spp --steps 1 --warmup 0 > $O/prof_cornell_$spp.json 2> $O/prof_cornell_$spp.txt
  grep "class 0" $O/prof_cornell_$spp.txt | tail -1 | cut -c1-700
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
