#!/bin/bash
out=gpurun_out/ad3; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $out/pytest.log
python tools/bench_as_shipped.py --scene materialtest
python tools/bench_as_shipped.py --scene materialtest --no-sobol
python tools/bench_as_shipped.py --scene materialtest --no-adaptive
python tools/bench_as_shipped.py --scene materialtest --no-adaptive --no-sobol
python tools/bench_as_shipped.py --scene cornell --spp 256 --spp-step 16
python tools/bench_as_shipped.py --scene cornell --spp 256 --spp-step 16 --no-sobol
python tools/bench_as_shipped.py --scene cornell --spp 256 --spp-step 16 --no-adaptive
python tools/bench_as_shipped.py --scene cornell --spp 256 --spp-step 16 --no-adaptive --no-sobol
