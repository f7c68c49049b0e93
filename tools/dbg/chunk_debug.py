"""Development aid: sample counts per pixel against chunk_samples and the scheduling options."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tungsten_amd as tg
import scenes
tmp = tempfile.mkdtemp()
path = scenes.materialtest(tmp, resolution=(192, 108), spp=8)
def render(**opts):
    r = tg.Renderer(path, seed=tg.DEFAULT_SEED)
    for k, v in opts.items():
        r.set_option(k, v)
    r.render()
    mean, ssum, count = r.image()
    c = r.counters()
    r.close()
    return mean, ssum, count, c
for opts in (dict(chunk_samples=1), dict(chunk_samples=1), dict(chunk_samples=1, streams=2), dict(chunk_samples=1, streams=1, blocks_per_cu=8),
             dict(chunk_samples=1, streams=4, blocks_per_cu=4), dict(chunk_samples=1, check_interval=1), dict(chunk_samples=1, check_interval=4),
             dict(chunk_samples=1, slots_per_block=64), dict(chunk_samples=1, max_slots=65536), dict(chunk_samples=1, wide_bvh=0),
             dict(chunk_samples=4), dict(chunk_samples=4), dict(chunk_samples=4, max_slots=131072*2)):
    m, s, cnt, c = render(**opts)
    bad = np.argwhere(cnt != 8)
    print(opts, "samples", c.samples, "iterations", c.iterations, "bad pixels", len(bad), bad[:4].tolist(), cnt[cnt != 8][:4].tolist())
