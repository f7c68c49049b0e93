export TGHIP_VERBOSE=1
bash tools/gpu_tune.sh r2b materialtest 64 "max_slots=2097152"
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_x.so bash tools/gpu_tune.sh r2b materialtest 64 "max_slots=2097152" "max_slots=2097152 threads_shadow=320" "max_slots=2097152 threads_shadow=256"
grep tghip gpurun_out/r2b/err.log | sort | uniq -c
