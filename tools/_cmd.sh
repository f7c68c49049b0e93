#!/bin/bash
run() { timeout 300 python bench.py --scene materialtest --spp 64 --no-extra --no-cpu-baseline "$@" | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],{k:v['avg_us'] for k,v in d['kernels'].items()})"; }
run
run --opt threads_shade_simple=128
run --opt threads_shade_simple=256
run --opt threads_shade_complex=64
run --opt threads_shade_complex=192
run --opt threads_closest=384
run --opt threads_closest=256
run --opt max_slots=1048576
run --opt max_slots=4194304
