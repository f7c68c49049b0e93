import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes, tungsten_amd as tg
tmp = tempfile.mkdtemp()
W, H, SPP = int(os.environ.get("RW", "480")), int(os.environ.get("RH", "270")), int(os.environ.get("SPP", "96"))
path = scenes.instances10k(tmp, resolution=(W, H), spp=SPP)
imgs = {}
for mode in (3, 2, 1, 0):
    r = tg.Renderer(path)
    r.set_option("wide_closest", mode & 1)
    r.set_option("wide_shadow", mode >> 1)
    r.render()
    mean, ssum, count = r.image()
    c = r.counters()
    r.close()
    imgs[mode] = mean
    print("wide_bvh", mode, "mean", mean.mean(axis=(0, 1)), "count ok", (count == SPP).all(), "closest", c.closest_rays, "shadow", c.shadow_rays, "samples", c.samples)
for m in (3, 2, 1):
    print("mode", m, "vs 0: pixels differing > 2%:", (np.abs(imgs[m] - imgs[0]).max(axis=-1)/(np.abs(imgs[0]).max(axis=-1) + 1e-2) > 0.02).mean())
d = np.abs(imgs[3] - imgs[0]).max(axis=-1)/(np.abs(imgs[0]).max(axis=-1) + 1e-2)
print("pixels differing > 2%:", (d > 0.02).mean(), "max", d.max())
ys, xs = np.where(d > 0.02)
print("rows of differing pixels (hist over 10 bands):", np.histogram(ys, bins=10, range=(0, H))[0], "cols:", np.histogram(xs, bins=10, range=(0, W))[0])
