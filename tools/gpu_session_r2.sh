#!/bin/bash
# Round-2 evidence session: GPU suite (+ per-sample divergence table), smoke, the default bench line, per-scene bench lines with
# rocprofv3 kernel statistics, HBM traffic and SQ counter passes, emulated tile-shard scaling.  Usage: tools/gpu_session_r2.sh <tag>
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== pytest -m gpu"
rm -f $out/diverge.jsonl
TG_DIVERGE_TABLE=$PWD/$out/diverge.jsonl timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $out/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== default bench line"
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc=$?"; cut -c1-400 $out/bench_default.json
for scene in materialtest mesh1m instances10k cornell; do
  spp=256; [ $scene = mesh1m ] && spp=32; [ $scene = instances10k ] && spp=32
  echo "== bench $scene"
  timeout 600 python bench.py --scene $scene --spp $spp --no-extra --no-cpu-baseline > $out/bench_$scene.json 2> $out/bench_$scene.err; echo "rc=$?"; cut -c1-300 $out/bench_$scene.json
  if [ $scene != cornell ]; then
    timeout 600 python bench.py --scene $scene --spp $spp --no-extra --no-cpu-baseline --no-traffic --opt wide_bvh=0 > $out/bench_${scene}_bvh2.json 2> $out/bench_${scene}_bvh2.err; echo "bvh2 rc=$?"; cut -c1-200 $out/bench_${scene}_bvh2.json
  fi
  echo "== rocprof stats $scene"
  pspp=$(( spp / 4 )); [ $scene = cornell ] && pspp=$spp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$scene -o stats -- python bench.py --scene $scene --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing --no-traffic > $out/prof_$scene.log 2>&1; echo "rc=$?"
  f=$(find $out/prof_$scene -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -v "at::native\|rocclr" $f > $out/${scene}_kernel_stats.csv && head -7 $out/${scene}_kernel_stats.csv
  rm -rf $out/prof_$scene
  echo "== SQ counters $scene"
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc_${scene}_SQ -o pmc -- python bench.py --scene $scene --spp $pspp --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-kernel-timing --no-traffic > $out/pmc_${scene}_SQ.log 2>&1; echo "SQ rc=$?"
  fs=$(find $out/pmc_${scene}_SQ -name '*counter_collection.csv' | head -1)
  [ -n "$fs" ] && python tools/pmc_sq.py $scene $fs $out/sq_counters.json > /dev/null
  rm -rf $out/pmc_${scene}_SQ
done
echo "== instances10k at BASELINE configs[4]'s own size"
timeout 300 python bench.py --scene instances10k --res 3840x2160 --spp 16 --no-extra --no-cpu-baseline --no-traffic > $out/bench_instances10k_4k.json 2> $out/bench_instances10k_4k.err; echo "rc=$?"; cut -c1-200 $out/bench_instances10k_4k.json
echo "== emulated tile-shard scaling (shard 0 of N on one GPU, no reduce)"
for scene in materialtest cornell; do
  for n in 1 2 4 8; do
    timeout 300 python bench.py --scene $scene --spp 256 --no-extra --no-cpu-baseline --no-kernel-timing --no-traffic --emulate-shards $n | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$scene shard 1/$n',d['ms_per_step'])"
  done
done | tee $out/emulated_scaling.txt
