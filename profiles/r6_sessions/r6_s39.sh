#!/bin/bash
# round 6, session 39: the equirectangular camera -- parity (goldens per sample, reference program with the plugin), and the headline with / without it (the camera code sits in
# nextPath's EXT variant, which k_finish_trace_closest_wide inlines: +250 instructions, +16 B of scratch there): base = the library of commit 6420fa5, alternated
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s39; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_samples.py tests/test_ref_binding.py -m gpu -q -k "equirect or thinlens or cornell_sobol or cornell_box_filter" > $O/gpu_equirect.txt 2>&1
tail -4 $O/gpu_equirect.txt
for round in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_base.so; else unset TUNGSTEN_AMD_LIB; fi
    timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 12 > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err
    timeout 600 python bench.py --scene mesh1m --no-cpu-baseline --no-extra --no-traffic --no-exclusive --no-clock --steps 4 > $O/mesh_${v}_$round.json 2> $O/mesh_${v}_$round.err
  done
done
unset TUNGSTEN_AMD_LIB
python - <<'PY'
import json
for sc in ("bench", "mesh"):
    for v in ("base", "new"):
        vals = []
        for r in (1, 2, 3):
            d = json.load(open("gpurun_out/r6_s39/%s_%s_%d.json" % (sc, v, r)))
            vals.append((d["value"], d["kernels"]["k_trace_closest"]["avg_us"]))
        print(sc, v, vals)
PY
