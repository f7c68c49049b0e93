#!/bin/bash
# round 6, session 38: the driver's N-rank command line rehearsed on the box's one GPU with the final binaries (TG_BENCH_SHARE_DEVICE=1: both ranks on device 0, gloo exchange),
# 2 and 4 ranks; and the one-process N-context path (--in-process) with two contexts on the one device
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s38; mkdir -p $O
for n in 2 4; do
  TG_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 5 --warmup 2 > $O/ranks_$n.json 2> $O/ranks_$n.err
  tail -c 400 $O/ranks_$n.err | tail -3
done
python - <<'PY'
import json
for n in (2, 4):
    try:
        d = json.loads(open("gpurun_out/r6_s38/ranks_%d.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["n_gpus"], d["result_ok"], d["config"]["parallelism"][:100], d.get("per_rank", {}).get("total_ms_per_step"))
    except Exception as e:
        print(n, "FAILED", e)
PY
