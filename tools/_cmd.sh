for i in 1 2; do
bash tools/gpu_tune.sh r1s materialtest 64 "max_slots=2097152" "max_slots=2097152 pool_pad=0" "max_slots=2097152 pool_pad=4352"
bash tools/gpu_tune.sh r1s cornell 256 "max_slots=2097152" "max_slots=2097152 pool_pad=0" "max_slots=2097152 pool_pad=4352"
done
