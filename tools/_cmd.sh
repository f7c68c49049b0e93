python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15
bash tools/gpu_tune.sh r1c cornell 256 "max_slots=1048576" "max_slots=524288" "max_slots=2097152" "max_slots=1048576 chunk_samples=8" "max_slots=1048576 blocks_per_cu=8" "max_slots=1048576 blocks_per_cu=2" "max_slots=1048576 check_interval=8"
bash tools/gpu_tune.sh r1c materialtest 64 "max_slots=1048576" "max_slots=524288" "max_slots=2097152" "max_slots=1048576 blocks_per_cu=7" "max_slots=1048576 blocks_per_cu=2"
