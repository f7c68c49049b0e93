import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes, oracle_lib, tungsten_amd as tg
tmp = tempfile.mkdtemp()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nlat = int(sys.argv[2]) if len(sys.argv) > 2 else 100
path = scenes.instances10k(tmp, resolution=(int(os.environ.get("RW", "160")), int(os.environ.get("RH", "90"))), spp=int(os.environ.get("SPP", "2")), count=count, n_lat=nlat, n_lon=nlat)
flat = tg.FlattenedScene(path)
d = flat.desc.contents
print("instances", d.num_instances, "wide", d.num_wide_nodes)
rs = np.random.RandomState(5)
n = 20000
lo, hi = np.array(list(d.bounds_lo)), np.array(list(d.bounds_hi))
o = lo + (hi - lo)*rs.rand(n, 3)*1.2 - 0.1*(hi - lo)
dirs = rs.randn(n, 3); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
rays = np.concatenate([o, np.full((n, 1), 1e-4), dirs, np.full((n, 1), np.inf)], axis=1).astype(np.float32)
oh, on, op = oracle_lib.trace_rays(flat.desc, rays, wide=True)
bh = oracle_lib.trace_rays(flat.desc, rays)[0]
r = tg.Renderer(path)
r.set_option("count_traversal", 1); r.reset_counters()
gh, _ = r.trace_rays(rays)
c = r.counters()
print("trace_rays: device vs oracle-wide same rec", (gh["rec"] == oh["rec"]).mean(), "vs bvh2", (gh["rec"] == bh["rec"]).mean(), "counts", c.nodes_visited, on, c.prims_tested, op)
SPP = int(os.environ.get("SPP", "2"))
sw = r.trace_samples(0, SPP)
r.close()
r = tg.Renderer(path)
r.set_option("wide_bvh", 0)
sb = r.trace_samples(0, SPP)
r.close()
diff = np.abs(sw - sb).max(axis=-1) > 1e-3*(np.abs(sb).max(axis=-1) + 1e-3)
print("per-sample wide vs bvh2 kernels: differing", diff.mean(), "by sample index", diff.mean(axis=(0, 1)), "means", sw.mean(axis=(0, 1, 2)), sb.mean(axis=(0, 1, 2)))
ys, xs, ss = np.where(diff)
for k in range(min(8, len(ys))):
    y, x, s_ = ys[k], xs[k], ss[k]
    osamp = oracle_lib.trace_sample(flat.desc, tg.DEFAULT_SEED, int(x), int(y), int(s_))
    print("  px", x, y, "s", s_, "wide", sw[y, x, s_], "bvh2", sb[y, x, s_], "oracle", osamp)
flat.close()
