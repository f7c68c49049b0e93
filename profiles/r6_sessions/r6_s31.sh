#!/bin/bash
# round 6, session 31: AtmosphericMedium on the device again (upload validation fixed), the whole media / ref-binding selection; the sustained shader
# clock during bench.py's renders read from sysfs every 50 ms (pp_dpm_sclk's current level) next to rocm-smi
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s31; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_samples.py tests/test_ref_binding.py tests/test_media.py -m gpu -q -k "atmosphere or fog or smoke" > $O/gpu_atmosphere.txt 2>&1
tail -15 $O/gpu_atmosphere.txt
ls /sys/class/drm/*/device/pp_dpm_sclk > $O/sysfs_nodes.txt 2>&1
( while true; do for f in /sys/class/drm/card*/device/pp_dpm_sclk; do grep '\*' $f 2>/dev/null | tr '\n' ' '; done; echo; sleep 0.05; done ) > $O/sclk_sysfs.txt 2>&1 &
POLL=$!
( while true; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | tr '\n' ' '; echo; done ) > $O/sclk_rocm_smi.txt 2>&1 &
POLL2=$!
timeout 600 python bench.py --steps 40 --no-cpu-baseline --no-extra --no-traffic > $O/bench_default.json 2> $O/bench_default.err
kill $POLL $POLL2
python - <<'PY'
import json, re
d = json.load(open("gpurun_out/r6_s31/bench_default.json"))
print(d["value"], d["result_ok"], d["ms_per_step"])
for name in ("sclk_sysfs.txt", "sclk_rocm_smi.txt"):
    vals = [int(m) for line in open("gpurun_out/r6_s31/" + name) for m in re.findall(r"(\d+)Mhz", line, re.I)]
    busy = [v for v in vals if v > 1500]
    print(name, "samples", len(vals), "above 1500 MHz:", len(busy), "min", min(busy) if busy else None, "max", max(busy) if busy else None, "mean", round(sum(busy)/max(len(busy), 1), 1))
PY
head -3 $O/sclk_sysfs.txt
