python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -5
bash tools/gpu_tune.sh r1i cornell 256 "max_slots=1048576" "max_slots=2097152" "max_slots=524288"
bash tools/gpu_tune.sh r1i materialtest 64 "max_slots=1048576" "max_slots=2097152" "max_slots=4194304"
