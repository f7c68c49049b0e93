#!/bin/bash
# round 2, session as: final GPU suite on the committed code; BASELINE configs[2] (materialtest 1920x1080, dielectric / rough dielectric)
out=gpurun_out/r2as
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $out/pytest.log
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 3 --warmup 1"
for m in dielectric rough_dielectric; do
  timeout 200 $B --scene materialtest --material $m --res 1920x1080 --spp 64 > $out/bench_materialtest_${m}_1080p.json 2> $out/bench_${m}.err; echo "rc=$?"
  python -c "
import json;d=json.loads(open('$out/bench_materialtest_${m}_1080p.json').read());k=d['kernels'];print('$m', d['value'], d['ms_per_step'], d['config']['workload'][:80], ' '.join('%s %.0fus'%(n,k[n]['avg_us']) for n in k), d['result_ok'])"
done
