#!/bin/bash
# round 6, session 32: the media kernels with the exponential-only transmittance helpers (k_shade<MASK_MEDIA> 233 k -> 140 k instructions) and the
# atmospheric medium: every media golden per sample, the 64x atmosphere render against the reference's hashes, media throughput against the
# library of the previous commit (libtungsten_hip_base.so), bench.py's sustained_clock field
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s32; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_samples.py tests/test_gpu_scale.py tests/test_media.py tests/test_ref_binding.py -m gpu -q -k "atmosphere or fog or smoke or non_exponential or volumetric or caustic" > $O/gpu_media.txt 2>&1
tail -6 $O/gpu_media.txt
for round in 1 2; do
  TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_base.so TG_MEDIA_ATMOSPHERE=0 timeout 600 python tools/bench_media.py 64 > $O/media_base_$round.jsonl 2> $O/media_base_$round.err
  timeout 600 python tools/bench_media.py 64 > $O/media_new_$round.jsonl 2> $O/media_new_$round.err
done
python - <<'PY'
import json
for tag in ("base_1", "new_1", "base_2", "new_2"):
    rows = [json.loads(l) for l in open("gpurun_out/r6_s32/media_%s.jsonl" % tag)]
    print(tag, " ".join("%s %.0f" % (r["scene"].replace("cornell_", ""), r["msamples_per_s"]) for r in rows))
PY
timeout 600 python bench.py --no-cpu-baseline --no-extra --no-traffic > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_s32/bench_default.json"))
print(d["value"], d["result_ok"], d.get("sustained_clock"))
PY
