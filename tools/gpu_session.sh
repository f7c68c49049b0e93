#!/bin/bash
# One GPU-box session: parity tests, bench lines, rocprofv3 kernel stats + HBM counters.  Usage: tools/gpu_session.sh <tag>
tag=${1:-session}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $out/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== BASELINE configs[2]: materialtest 1920x1080 @ 1024 spp, rough conductor (shipped) and dielectric, one step each"
for mat in shipped dielectric; do
  timeout 600 python bench.py --scene materialtest --material $mat --res 1920x1080 --spp 1024 --steps 1 --warmup 0 --no-extra --no-cpu-baseline > $out/bench_c3_$mat.json 2> $out/bench_c3_$mat.err; echo "rc=$?"
  python -c "import json;d=json.loads(open('$out/bench_c3_$mat.json').read());print('$mat',d['value'],d['ms_per_step'],d['result_ok'],{k:(v['avg_us'],v['gbs']) for k,v in d['kernels'].items()})"
done
echo "== instances10k 1920x1080 @ 32 spp and 3840x2160 @ 16 spp"
timeout 600 python bench.py --scene instances10k --spp 32 --no-extra --cpu-seconds 8 > $out/bench_instances10k.json 2> $out/bench_instances10k.err; echo "rc=$?"; cut -c1-700 $out/bench_instances10k.json
timeout 600 python bench.py --scene instances10k --res 3840x2160 --spp 16 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $out/bench_instances10k_4k.json 2> $out/bench_instances10k_4k.err; echo "rc=$?"; cut -c1-300 $out/bench_instances10k_4k.json
echo "== strong-scaling emulation (shard 0 of N on one GPU)"
for scene in cornell materialtest; do
  spp=256; [ $scene = materialtest ] && spp=64
  for n in 1 2 4 8; do
    timeout 300 python bench.py --scene $scene --spp $spp --no-extra --no-cpu-baseline --no-kernel-timing --emulate-shards $n | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$scene shard 1/$n',d['ms_per_step'])"
  done
done | tee $out/emulated_scaling.txt
echo "== as shipped (Sobol + adaptive, passes of 16 spp)"
for args in "--scene materialtest" "--scene materialtest --no-sobol --no-adaptive" "--scene cornell --spp 256" "--scene cornell --spp 256 --no-sobol --no-adaptive"; do
  python tools/bench_as_shipped.py $args
done | tee $out/as_shipped.jsonl
for scene in cornell materialtest mesh1m; do
  spp=256; [ $scene = materialtest ] && spp=64; [ $scene = mesh1m ] && spp=32
  echo "== bench $scene"
  timeout 600 python bench.py --scene $scene --spp $spp --no-extra > $out/bench_$scene.json 2> $out/bench_$scene.err; echo "rc=$?"; cat $out/bench_$scene.json; tail -3 $out/bench_$scene.err
  echo "== rocprof stats $scene"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$scene -o stats -- python bench.py --scene $scene --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing > $out/prof_$scene.log 2>&1; echo "rc=$?"
  f=$(find $out/prof_$scene -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/${scene}_kernel_stats.csv && head -8 $f
  find $out/prof_$scene -name '*kernel_trace.csv' -delete; find $out/prof_$scene -name '*.db' -delete
  echo "== rocprof pmc $scene"
  pmcspp=$(( spp / 4 )); [ $scene = cornell ] && pmcspp=$spp   # the Cornell render is ONE launch: collect at the bench's spp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${scene}_$c -o pmc -- python bench.py --scene $scene --spp $pmcspp --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-kernel-timing > $out/pmc_${scene}_$c.log 2>&1; echo "$c rc=$?"
  done
  # where the waves spend their cycles (SQ block, its own pass)
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc_${scene}_SQ -o pmc -- python bench.py --scene $scene --spp $pmcspp --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-kernel-timing > $out/pmc_${scene}_SQ.log 2>&1; echo "SQ rc=$?"
  fs=$(find $out/pmc_${scene}_SQ -name '*counter_collection.csv' | head -1)
  [ -n "$fs" ] && python tools/pmc_sq.py $scene $fs $out/sq_counters.json
  ff=$(find $out/pmc_${scene}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
  fw=$(find $out/pmc_${scene}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  [ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $scene $ff $fw $out/traffic.json
  find $out -name '*counter_collection.csv' -delete; find $out -name '*.db' -delete
done
