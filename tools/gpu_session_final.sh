#!/bin/bash
# Trimmed end-of-round session: parity tests, smoke, the three bench lines with rocprofv3 kernel stats and HBM counters,
# BASELINE configs[2] and the as-shipped lines.  Usage: tools/gpu_session_final.sh <tag>
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $out/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for scene in cornell materialtest mesh1m; do
  spp=256; [ $scene = materialtest ] && spp=64; [ $scene = mesh1m ] && spp=32
  echo "== bench $scene"
  timeout 400 python bench.py --scene $scene --spp $spp --no-extra > $out/bench_$scene.json 2> $out/bench_$scene.err; echo "rc=$?"; cut -c1-900 $out/bench_$scene.json; tail -2 $out/bench_$scene.err
  echo "== rocprof stats $scene"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$scene -o stats -- python bench.py --scene $scene --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing > $out/prof_$scene.log 2>&1; echo "rc=$?"
  f=$(find $out/prof_$scene -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/${scene}_kernel_stats.csv && head -6 $f
  find $out/prof_$scene -name '*kernel_trace.csv' -delete; find $out/prof_$scene -name '*.db' -delete
  echo "== rocprof pmc $scene"
  pmcspp=$(( spp / 4 )); [ $scene = cornell ] && pmcspp=$spp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${scene}_$c -o pmc -- python bench.py --scene $scene --spp $pmcspp --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-kernel-timing > $out/pmc_${scene}_$c.log 2>&1; echo "$c rc=$?"
  done
  ff=$(find $out/pmc_${scene}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
  fw=$(find $out/pmc_${scene}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  [ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $scene $ff $fw $out/traffic.json
  find $out -name '*counter_collection.csv' -delete; find $out -name '*.db' -delete
done
echo "== BASELINE configs[2]"
for mat in shipped dielectric; do
  timeout 300 python bench.py --scene materialtest --material $mat --res 1920x1080 --spp 1024 --steps 1 --warmup 0 --no-extra --no-cpu-baseline > $out/bench_c3_$mat.json 2> $out/bench_c3_$mat.err; echo "rc=$?"
  python -c "import json;d=json.loads(open('$out/bench_c3_$mat.json').read());print('$mat',d['value'],d['ms_per_step'],d['result_ok'])"
done
echo "== as shipped"
for args in "--scene materialtest" "--scene cornell --spp 256"; do
  timeout 200 python tools/bench_as_shipped.py $args
done | tee $out/as_shipped.jsonl
timeout 120 python tools/bench_media.py 64 | tee $out/media_bench.jsonl
