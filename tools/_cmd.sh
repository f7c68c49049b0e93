python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6
bash tools/gpu_tune.sh r1z cornell 256 "max_slots=2097152 blocks_per_cu=2" "max_slots=2097152 blocks_per_cu=4" "max_slots=4194304 blocks_per_cu=8" "max_slots=2097152 blocks_per_cu=8 threads_shade_simple=128" "max_slots=2097152 blocks_per_cu=8 threads_shade_simple=192"
