import sys, os, tempfile, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes, tungsten_amd as tg
from tungsten_amd import capi
tmp = tempfile.mkdtemp()
W, H, SPP = 480, 270, 8
path = scenes.instances10k(tmp, resolution=(W, H), spp=SPP)
flat = tg.FlattenedScene(path)
def run(flags, wide_shadow, opts=()):
    ctx = tg.lib.tghip_create(0)
    tg.lib.tghip_set_option(ctx, b"wide_closest", 0)
    tg.lib.tghip_set_option(ctx, b"wide_shadow", wide_shadow)
    for k, v in opts:
        assert tg.lib.tghip_set_option(ctx, k, v) == 0
    assert tg.lib.tghip_upload_scene(ctx, flat.desc) == 0
    p = tg.TgHipPassDesc(0, SPP, tg.DEFAULT_SEED, 0, 1, flags)
    assert tg.lib.tghip_render_pass(ctx, C.byref(p)) == 0 and tg.lib.tghip_wait(ctx) == 0
    s, c = np.empty((W*H, 3), np.float32), np.empty(W*H, np.uint32)
    assert tg.lib.tghip_download_framebuffer(ctx, s.ctypes.data, c.ctypes.data, W*H) == 0
    tg.lib.tghip_destroy(ctx)
    return s
b = run(0, 0)
print("static-shadow mean", b.mean(axis=0)/SPP)
for opts in ((), ((b"count_traversal", 1),), ((b"leaf_batch", 2),), ((b"wide_closest", 1),), ((b"wide_closest", 1), (b"count_traversal", 1))):
    a = run(0, 1, opts)
    print("wide-shadow with", opts, "mean", a.mean(axis=0)/SPP, "max abs diff vs static", np.abs(a - b).max())
