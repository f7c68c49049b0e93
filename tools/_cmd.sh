#!/bin/bash
out=gpurun_out/inst1; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 $out/pytest.log
