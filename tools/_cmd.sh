#!/bin/bash
out=gpurun_out/inst2; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "many_instances or instances" > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $out/pytest.log
timeout 600 python bench.py --scene instances10k --spp 32 --no-extra --cpu-seconds 10 > $out/bench_instances10k.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err
python -c "
import json;d=json.loads(open('$out/bench_instances10k.json').read());print(d['value'],d['ms_per_step'],d['kernels'],d['cpu_baseline'],d['bvh'],d['rays_per_sample'],d['nodes_per_ray'],d['prims_per_ray'])"
