"""The N > 1 path on CPU: 2 ranks over gloo run the same tile sharding + framebuffer reduce bench.py uses on
GPUs (tungsten_amd/dist.py), with the oracle standing in for the device renderer.  Rank 0 must end up with
the unsharded image, bit for bit (disjoint tile ownership => exact reduction)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import tungsten_amd as tg
from tungsten_amd import dist as tgdist
import oracle_lib, scenes
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
tmp = sys.argv[2]
path = scenes.cornell(tmp, name="r%d.json" % rank, resolution=(80, 45), spp=2)
flat = tg.FlattenedScene(path)
p = tgdist.shard_pass(rank, world, 0, 2, 1234)
s, c = oracle_lib.render(flat.desc, 80, 45, p.spp_begin, p.spp_end, p.seed, shard_index=p.shard_index, shard_count=p.shard_count)
# every pixel this rank owns lies in one of its tiles, and nothing else was touched
owned = np.zeros((45, 80), bool)
for t in tgdist.owned_tiles(rank, world, 80, 45):
    ty, tx = divmod(t, 5)
    owned[ty*16:ty*16 + 16, tx*16:tx*16 + 16] = True
assert ((c > 0) == owned).all()
fs, fc = torch.from_numpy(s), torch.from_numpy(c.astype(np.int32))
tgdist.reduce_framebuffer(fs, fc, dst=0)
if rank == 0:
    ws, wc = oracle_lib.render(flat.desc, 80, 45, 0, 2, 1234)
    assert (fs.numpy() == ws).all() and (fc.numpy() == wc.astype(np.int32)).all()
    print("DIST_OK")
# the product's own exchange step (tghip_comm_* behind tgdist.init_rank_comm) cannot be built without a device: every rank must learn that, agree on
# it through the all-reduce and get the same reason back -- bench.py then falls back to reduce_framebuffer above, and says so
why = tgdist.init_rank_comm(tg.lib, None, rank, world)
assert why is not None and "failed on a rank" in why
dist.barrier()
dist.destroy_process_group()
'''


ADAPTIVE_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import tungsten_amd as tg
from tungsten_amd import dist as tgdist
import oracle_lib, scenes
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, SPP, STEP, SEED = 70, 42, 48, 16, 1234
path = scenes.cornell(sys.argv[2], name="a%d.json" % rank, resolution=(W, H), spp=SPP, spp_step=STEP,
                      renderer={"adaptive_sampling": True, "stratified_sampler": True})
flat = tg.FlattenedScene(path)
ssum = np.zeros((H, W, 3), np.float32); count = np.zeros((H, W), np.uint32)
records = np.zeros(((W + 3)//4)*((H + 3)//4), oracle_lib.DEVICE_RECORD_DTYPE)
aux = np.zeros((H, W), oracle_lib.AUX_DTYPE)
import ctypes as C
def render_pass(p):      # the oracle stands in for tghip_render_pass + tghip_wait on this rank's device
    rc = oracle_lib._lib.oracle_render_aux(flat.desc, p, C.c_void_p(ssum.ctypes.data), C.c_void_p(count.ctypes.data), C.c_void_p(records.ctypes.data),
                                           C.c_void_p(aux.ctypes.data), None, 2)
    assert rc == 0
sch = tgdist.render_loop(render_pass, lambda: records, W, H, SPP, STEP, SEED, rank=rank, world=world, adaptive=True, sobol=True,
                         output_buffers=True)
merged_aux = tgdist.reduce_output_buffers(aux, dst=0)
fs, fc = torch.from_numpy(ssum), torch.from_numpy(count.astype(np.int32))
tgdist.reduce_framebuffer(fs, fc, dst=0)
if rank == 0:
    ws, wc, wrec, _, waux = oracle_lib.integrate_aux(flat.desc, W, H, SEED, SPP, STEP, True, True)
    assert merged_aux.tobytes() == waux.tobytes()      # the output buffers of both ranks' tiles, bit for bit
    assert (merged_aux["count"][..., 0] == wc).all()
    final = sch.records.reshape(wrec[-1].shape)
    for f in ("sample_count", "next_sample_count", "sample_index", "mean", "running_variance"):
        assert (final[f] == wrec[-1][f]).all(), f
    assert (fc.numpy() == wc.astype(np.int32)).all() and (fs.numpy() == ws).all()
    assert int(wc.max()) > int(wc.min())        # the schedule really was adaptive
    print("ADAPTIVE_DIST_OK")
dist.barrier()
dist.destroy_process_group()
'''


def _run_two_ranks(tmp_path, source, token):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path/"worker.py"
    script.write_text(source)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert token in outs[0]


def test_two_rank_adaptive_pass_loop(tmp_path):
    """Adaptive sampling + Sobol' across 2 ranks: each rank renders its tiles of every pass, the ranks exchange the
    SampleRecords (tungsten_amd/dist.py: merge_records) and run the same scheduler; records, sample counts and framebuffer
    must equal the single-process loop's bit for bit -- and so must the auxiliary output buffers (TGHIP_PASS_AUX passes,
    tungsten_amd/dist.py: reduce_output_buffers)."""
    _run_two_ranks(tmp_path, ADAPTIVE_WORKER, "ADAPTIVE_DIST_OK")


def test_two_rank_tile_shard_and_reduce(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path/"worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0]


def test_shard_pass_validation():
    from tungsten_amd import dist as tgdist
    import pytest
    p = tgdist.shard_pass(3, 8, 0, 256, 0xBA5EBA11)
    assert (p.shard_index, p.shard_count, p.spp_end) == (3, 8, 256)
    with pytest.raises(ValueError):
        tgdist.shard_pass(8, 8, 0, 1, 0)
    tiles = [t for r in range(8) for t in tgdist.owned_tiles(r, 8, 1280, 720)]
    assert sorted(tiles) == list(range(80*45))


def test_tile_ownership_is_a_two_dimensional_interleave():
    """tghip_tile_owner (include/tungsten_hip.h) as mirrored by dist.tile_owner: every tile has one owner, the shards are
    balanced to within one tile per tile row, and -- the point of the diagonal deal -- no shard is a set of vertical stripes,
    for the BASELINE resolutions (whose tiles per row are all multiples of 8) and every world size up to 16."""
    from tungsten_amd import dist as tgdist
    for (w, h) in ((1280, 720), (1920, 1080), (3840, 2160), (80, 45), (333, 211)):
        tx_n, ty_n = (w + 15)//16, (h + 15)//16
        for world in range(1, 17):
            lists = [tgdist.owned_tiles(r, world, w, h) for r in range(world)]
            assert sorted(t for l in lists for t in l) == list(range(tx_n*ty_n))
            assert all(l == sorted(l) for l in lists)                       # rendered in row-major order
            sizes = [len(l) for l in lists]
            assert max(sizes) - min(sizes) <= ty_n
            if world > 1 and tx_n >= world and ty_n >= world:
                for l in lists:
                    cols = {t % tx_n for t in l}
                    rows = {t//tx_n for t in l}
                    assert len(cols) == tx_n and len(rows) == ty_n          # present in every tile column and every tile row
