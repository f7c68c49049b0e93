#!/bin/bash
# round 6, session 41: the folded finish of flag-less passes through nextPath's lean variant (k_finish_trace_closest_wide<., ., false>: 4 220 instead of 6 436 instructions,
# no thin-lens / Sobol' / record / auxiliary code) -- option finish_lean 0 / 1 alternated on the metric's workload and on mesh1m; parity of the per-sample goldens and scale hashes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s41; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_samples.py tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q -k "materialtest or mesh1m or scheduling or water" > $O/gpu_parity.txt 2>&1
tail -4 $O/gpu_parity.txt
for round in 1 2 3; do
  for v in 0 1; do
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 12 --opt finish_lean=$v > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err
    timeout 600 python bench.py --scene mesh1m --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 4 --opt finish_lean=$v > $O/mesh_${v}_$round.json 2> $O/mesh_${v}_$round.err
  done
done
python - <<'PY'
import json
for sc in ("bench", "mesh"):
    for v in (0, 1):
        vals = []
        for r in (1, 2, 3):
            d = json.load(open("gpurun_out/r6_s41/%s_%d_%d.json" % (sc, v, r)))
            vals.append((d["value"], d["kernels"]["k_trace_closest"]["avg_us"], d["image_mean"][0]))
        print(sc, "finish_lean", v, vals)
PY
