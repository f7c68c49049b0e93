bash tools/gpu_tune.sh r2d materialtest 64 "leaf_batch=1" "leaf_batch=4" "leaf_batch=8" "leaf_batch=12" "leaf_batch=20" "leaf_batch=32" "leaf_batch=1"
bash tools/gpu_tune.sh r2d mesh1m 32 "leaf_batch=1" "leaf_batch=8"
