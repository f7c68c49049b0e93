#!/bin/bash
# round 6, session 33: the homogeneous medium's exponential-transmittance fast path (run-time branch) against the previous commit's library and against a
# diagnostic build without the eight other transmittances (PT_EXP_TRANS_ONLY: 47 k instructions) -- media parity first
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s33; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_samples.py tests/test_gpu_scale.py tests/test_media.py tests/test_ref_binding.py -m gpu -q -k "atmosphere or fog or smoke or non_exponential or volumetric or caustic" > $O/gpu_media.txt 2>&1
tail -4 $O/gpu_media.txt
for round in 1 2; do
  for v in base texp; do
    TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_$v.so TG_MEDIA_ATMOSPHERE=0 timeout 600 python tools/bench_media.py 64 > $O/media_${v}_$round.jsonl 2> $O/media_${v}_$round.err
  done
  TG_MEDIA_ATMOSPHERE=0 timeout 600 python tools/bench_media.py 64 > $O/media_new_$round.jsonl 2> $O/media_new_$round.err
done
python - <<'PY'
import json
for tag in ("base_1", "new_1", "texp_1", "base_2", "new_2", "texp_2"):
    rows = [json.loads(l) for l in open("gpurun_out/r6_s33/media_%s.jsonl" % tag)]
    print("%-7s" % tag, " ".join("%s %.0f" % (r["scene"].replace("cornell_", ""), r["msamples_per_s"]) for r in rows))
PY
