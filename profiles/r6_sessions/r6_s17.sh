#!/bin/bash
# round 6, session 17: k_shade's residency throttled by unused LDS (does a SIMD that holds two shading waves instead of three take a walk's wave next to them?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_s17; mkdir -p $O
timeout 900 python tools/sweep.py --scene materialtest --steps 4 --repeat 2 -- "shade_lds_pad=0" "shade_lds_pad=16384" "shade_lds_pad=28672" "shade_lds_pad=57344" > $O/sweep.jsonl 2> $O/sweep.err
cut -c1-250 $O/sweep.jsonl
timeout 600 python tools/sweep.py --scene mesh1m --steps 3 -- "shade_lds_pad=0" "shade_lds_pad=28672" "shade_lds_pad=57344" > $O/sweep_mesh1m.jsonl 2> $O/sweep_mesh1m.err
cut -c1-250 $O/sweep_mesh1m.jsonl
