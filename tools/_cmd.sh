python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -8
bash tools/gpu_tune.sh r1w cornell 256 "max_slots=2097152"
bash tools/gpu_tune.sh r1w materialtest 64 "max_slots=2097152"
