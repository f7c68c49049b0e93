#!/bin/bash
# round 6, session 34: glibc's double exp / log / erf restated on the device (pt_libm.h: expD / logD / erfD) against the host libm, and the atmospheric medium on them
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s34; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_libm.py -m gpu -q > $O/gpu_libm.txt 2>&1
tail -8 $O/gpu_libm.txt
timeout 1200 python -m pytest tests/test_gpu_samples.py tests/test_gpu_scale.py tests/test_media.py tests/test_ref_binding.py -m gpu -q -k "atmosphere or fog or smoke" > $O/gpu_media.txt 2>&1
tail -4 $O/gpu_media.txt
TG_MEDIA_ATMOSPHERE=1 timeout 600 python tools/bench_media.py 64 > $O/media_new.jsonl 2> $O/media_new.err
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r6_s34/media_new.jsonl")]
print(" ".join("%s %.0f" % (r["scene"].replace("cornell_", ""), r["msamples_per_s"]) for r in rows))
PY
