"""Ad-hoc GPU probe used during development: render a scene variant, print throughput + counters."""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tungsten_amd as tg
import scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell")
ap.add_argument("--res", default="1280x720")
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--step", type=int, default=0)
ap.add_argument("--slots", type=int, default=0)
ap.add_argument("--count", type=int, default=0)
ap.add_argument("--check", type=int, default=0)
ap.add_argument("--bpc", type=int, default=0)
ap.add_argument("--save", default="")
a = ap.parse_args()
w, h = [int(v) for v in a.res.split("x")]
tmp = tempfile.mkdtemp()
mk = scenes.cornell if a.scene == "cornell" else scenes.materialtest
path = mk(tmp, resolution=(w, h), spp=a.spp, spp_step=a.step or a.spp)
t0 = time.time()
r = tg.Renderer(path)
print("open %.3fs  nodes %d recs %d depth %d" % (time.time() - t0, r.info.num_nodes, r.info.num_recs, r.info.bvh_depth))
if a.slots: r.set_option("max_slots", a.slots)
if a.count: r.set_option("count_traversal", 1)
if a.check: r.set_option("check_interval", a.check)
if a.bpc: r.set_option("blocks_per_cu", a.bpc)
secs = r.render()
c = r.counters()
mean, ssum, count = r.image()
n = w*h*a.spp
print("render %.4fs  %.2f Msamples/s  kernel_ms %.2f  iterations %d" % (secs, n/secs*1e-6, c.ms_total, c.iterations))
print("samples %d closest %d shadow %d rays/sample %.2f nodes %d prims %d" % (c.samples, c.closest_rays, c.shadow_rays,
      (c.closest_rays + c.shadow_rays)/max(c.samples, 1), c.nodes_visited, c.prims_tested))
print("mean rgb", mean.mean(axis=(0, 1)), "count min/max", count.min(), count.max(), "nan", np.isnan(mean).sum())
if a.save:
    tg.lib.tgh_save_pfm(a.save.encode(), mean.ctypes.data, w, h)
