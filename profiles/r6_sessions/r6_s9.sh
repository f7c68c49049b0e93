#!/bin/bash
# round 6, session 9: k_trace_closest_instw, phase T = (node step, leaves, pops) x inst_tree_steps -- parity, sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_samples.py tests/test_gpu_scale.py -m gpu -q -k "instance or instanced or instances" > $O/gpu_instances.txt 2>&1
tail -5 $O/gpu_instances.txt
B="inst_wide=1,inst_refill_at=48,inst_tight=0"
timeout 1500 python tools/sweep.py --scene instances10k --steps 2 -- "inst_wide=0" "$B,inst_phase_min=24,inst_tree_steps=1" "$B,inst_phase_min=24,inst_tree_steps=2" "$B,inst_phase_min=24,inst_tree_steps=3" "$B,inst_phase_min=24,inst_tree_steps=4" "$B,inst_phase_min=32,inst_tree_steps=2" "$B,inst_phase_min=64,inst_tree_steps=2" "$B,inst_phase_min=24,inst_tree_steps=2,leaf_batch_bvh2=8" "$B,inst_phase_min=24,inst_tree_steps=2,leaf_batch_bvh2=24" "inst_wide=1,inst_refill_at=32,inst_tight=0,inst_phase_min=24,inst_tree_steps=2" "inst_wide=1,inst_refill_at=56,inst_tight=0,inst_phase_min=24,inst_tree_steps=2" "$B,inst_phase_min=24,inst_tree_steps=1" "inst_wide=0" > $O/sweep.jsonl 2> $O/sweep.err
cat $O/sweep.jsonl | cut -c1-230
timeout 600 python bench.py --scene instances10k --no-cpu-baseline --no-extra --no-traffic > $O/bench_instances10k.json 2> $O/bench_instances10k.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_s9/bench_instances10k.json"))
print(d["value"], d["result_ok"], d["nodes_per_ray"], {n: k["avg_us"] for n, k in d["kernels"].items()})
print(json.dumps(d.get("walk", {}).get("closest_hit")))
PY
