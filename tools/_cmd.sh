python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4
bash tools/gpu_tune.sh r2a materialtest 64 "max_slots=2097152" "max_slots=2097152"
bash tools/gpu_tune.sh r2a mesh1m 32 "max_slots=2097152"
