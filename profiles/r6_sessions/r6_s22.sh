#!/bin/bash
# round 6, session 22: BASELINE configs[2..3] at their stated resolutions and spp on the round's binaries
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s22; mkdir -p $O
Q="--no-cpu-baseline --no-extra --no-exclusive --steps 2 --warmup 1"
timeout 900 python bench.py $Q --scene materialtest --material dielectric --res 1920x1080 --spp 1024 --count-spp 32 > $O/bench_c3_dielectric_1080p_1024spp.json 2> $O/c3a.err
timeout 900 python bench.py $Q --scene materialtest --material rough_dielectric --res 1920x1080 --spp 1024 --count-spp 32 > $O/bench_c3_rough_dielectric_1080p_1024spp.json 2> $O/c3b.err
timeout 900 python bench.py $Q --scene mesh1m --res 1920x1080 --spp 512 --count-spp 32 > $O/bench_c4_mesh1m_1080p_512spp.json 2> $O/c4.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r6_s22/*.json")):
    try:
        d = json.load(open(f)); print("%-52s %8.2f Msamples/s ok %s" % (os.path.basename(f), d["value"], d["result_ok"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
