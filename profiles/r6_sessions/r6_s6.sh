#!/bin/bash
# round 6, session 6: k_trace_closest_instw (instanced scenes: masters through the wide BVH, phase vote) -- parity, then the A/B and the vote's sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py tests/test_gpu_scale.py -m gpu -q -x -k "instance or instanced or instances" > $O/gpu_instances.txt 2>&1
tail -5 $O/gpu_instances.txt
timeout 900 python tools/sweep.py --scene instances10k --steps 2 --repeat 2 -- "inst_wide=0,inst_phase_min=16,inst_refill_at=48" "inst_wide=1,inst_phase_min=16,inst_refill_at=48" > $O/sweep_ab.jsonl 2> $O/sweep_ab.err
cat $O/sweep_ab.jsonl
timeout 1200 python tools/sweep.py --scene instances10k --steps 2 -- "inst_wide=1,inst_phase_min=1,inst_refill_at=48" "inst_wide=1,inst_phase_min=8,inst_refill_at=48" "inst_wide=1,inst_phase_min=24,inst_refill_at=48" "inst_wide=1,inst_phase_min=32,inst_refill_at=48" "inst_wide=1,inst_phase_min=64,inst_refill_at=48" "inst_wide=1,inst_phase_min=16,inst_refill_at=32" "inst_wide=1,inst_phase_min=16,inst_refill_at=40" "inst_wide=1,inst_phase_min=16,inst_refill_at=56" "inst_wide=1,inst_phase_min=24,inst_refill_at=56,leaf_batch_bvh2=8" "inst_wide=1,inst_phase_min=24,inst_refill_at=56,leaf_batch_bvh2=24" > $O/sweep_vote.jsonl 2> $O/sweep_vote.err
cat $O/sweep_vote.jsonl
for v in 0 1; do
  timeout 600 python bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic --opt inst_wide=$v > $O/bench_instances10k_wide$v.json 2> $O/bench_instances10k_wide$v.err
done
python - <<'PY'
import json
for v in (0, 1):
    try:
        d = json.load(open("gpurun_out/r6_s6/bench_instances10k_wide%d.json" % v))
        print(v, d["value"], d["result_ok"], d["nodes_per_ray"], {n: k["avg_us"] for n, k in d["kernels"].items()})
        print(json.dumps(d.get("walk", {}).get("closest_hit")))
    except Exception as e:
        print(v, "FAILED", e)
PY
