#!/bin/bash
# round 6, session 30: AtmosphericMedium on the device (goldens per sample, the reference program with the plugin), the media suite; the sustained
# shader clock while bench.py's timed region runs (rocm-smi polled every 0.25 s)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_samples.py tests/test_ref_binding.py tests/test_media.py -m gpu -q -k "atmosphere or fog or smoke" > $O/gpu_atmosphere.txt 2>&1
tail -15 $O/gpu_atmosphere.txt
rocm-smi --showclocks > $O/clocks_idle.txt 2>&1
( while true; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/clocks_during_bench.txt 2>&1 &
POLL=$!
timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic > $O/bench_default.json 2> $O/bench_default.err
kill $POLL
python - <<'PY'
import json, re
d = json.load(open("gpurun_out/r6_s30/bench_default.json"))
print(d["value"], d["result_ok"])
vals = []
for line in open("gpurun_out/r6_s30/clocks_during_bench.txt"):
    m = re.search(r"sclk[^(]*\((\d+)Mhz\)", line, re.I)
    if m: vals.append(int(m.group(1)))
print("sclk samples", len(vals), "min", min(vals) if vals else None, "max", max(vals) if vals else None, "mean", sum(vals)/max(len(vals), 1))
PY
head -3 $O/clocks_during_bench.txt
