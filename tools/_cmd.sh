export TGHIP_VERBOSE=1
python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
bash tools/gpu_tune.sh r1p materialtest 64 "max_slots=2097152 threads_closest=320" "max_slots=2097152 threads_closest=320 threads_shadow=192" "max_slots=2097152 threads_closest=320 threads_shadow=320" "max_slots=2097152 threads_closest=256" "max_slots=2097152 threads_closest=192" "max_slots=2097152 threads_closest=320"
grep tghip gpurun_out/r1p/err.log | sort | uniq -c
