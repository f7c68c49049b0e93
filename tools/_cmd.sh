#!/bin/bash
out=gpurun_out/resume1; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $out/pytest.log
cd /tmp && cp -r $GRAFT_REPO_ROOT/scenes/cornell-box /tmp/cb && python - <<'PY'
import json
s=json.load(open('/tmp/cb/scene.json')); s['renderer'].update(spp=64, spp_step=16, enable_resume_render=True, resume_render_file='/tmp/cb/state.dat', output_file='/tmp/cb/o.png', hdr_output_file='/tmp/cb/o.pfm')
s['camera']['resolution']=[320,180]
json.dump(s, open('/tmp/cb/s.json','w'))
PY
$GRAFT_REPO_ROOT/tungsten_amd/lib/tungsten_hip /tmp/cb/s.json | tail -4
$GRAFT_REPO_ROOT/tungsten_amd/lib/tungsten_hip /tmp/cb/s.json | tail -4
