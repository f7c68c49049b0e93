bash tools/gpu_tune.sh r1v materialtest 64 "max_slots=2097152" "max_slots=2097152"
TUNGSTEN_AMD_LIB=$PWD/tungsten_amd/lib/libtungsten_hip_s2.so bash tools/gpu_tune.sh r1v materialtest 64 "max_slots=2097152" "max_slots=2097152 threads_shade_simple=128" "max_slots=2097152"
bash tools/gpu_tune.sh r1v cornell 256 "max_slots=2097152"
