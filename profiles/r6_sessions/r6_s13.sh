#!/bin/bash
# round 6, session 13: evidence on the round's binaries -- GPU suite, the driver's bench line, the other scenes, rocprofv3 kernel stats, counters of the
# instanced walk, emulated shards, BASELINE configs[4] at its resolution
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s13; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
tail -3 $O/gpu_suite.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
for sc in cornell mesh1m instances10k; do
  timeout 700 python bench.py --scene $sc --no-cpu-baseline --no-extra > $O/bench_$sc.json 2> $O/bench_$sc.err
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_mt -o mt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --steps 2 > $GRAFT_REPO_ROOT/$O/prof_mt.json 2> $GRAFT_REPO_ROOT/$O/prof_mt.err )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_inst -o inst -- python $GRAFT_REPO_ROOT/bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --no-exclusive --steps 2 > $GRAFT_REPO_ROOT/$O/prof_inst.json 2> $GRAFT_REPO_ROOT/$O/prof_inst.err )
find $O/prof_mt $O/prof_inst -name "*kernel_stats*" | head
find $O/prof_mt $O/prof_inst -type f ! -name "*stats*" -delete 2>/dev/null
timeout 900 python tools/pmc_variants.py --out $O/sq_counters_instances10k.json --scene instances10k --spp 32 --groups lane,sq,tcp,tcc --timeout 150 > $O/sq_counters_instances10k.txt 2>&1
for n in 2 4 8; do
  timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --emulate-shards $n > $O/emulated_shards$n.json 2> $O/emulated_shards$n.err
done
timeout 900 python bench.py --scene instances10k --res 3840x2160 --spp 16 --steps 2 --no-cpu-baseline --no-extra --no-exclusive --count-spp 4 > $O/bench_c5_instances10k_4k_16spp.json 2> $O/bench_c5.err
timeout 600 python tools/bench_as_shipped.py --repeats 3 > $O/as_shipped.json 2> $O/as_shipped.err
ls -la $O
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_s13/*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d.get("value", d.get("msamples_per_s")), d.get("result_ok"), (d.get("extra") or {}) and {k: v.get("value") for k, v in d["extra"].items() if isinstance(v, dict)}, d.get("emulated_shards", ""))
    except Exception as e:
        print(f, "FAILED", str(e)[:80])
PY
