"""Multi-GPU decomposition of a render pass: one process per GPU, 16x16 tiles (the reference's
own dicing, integrators/path_tracer/PathTraceIntegrator.cpp:27-42) dealt to ranks along diagonals (tile_owner), and ONE exchange
step -- the sum-reduce of the float framebuffer (+ sample counts) to the root over RCCL/xGMI
(torch.distributed backend "nccl"; "gloo" in the CPU tests).  It is the in-process equivalent of the reference's
manual `hdrmanip --merge` of independently rendered images (src/hdrmanip/hdrmanip.cpp:69-112).

Tile ownership is disjoint, so every pixel receives exactly one non-zero addend: the reduction is exact and
independent of the collective's ring order."""
from . import capi


def shard_pass(rank, world, spp_begin, spp_end, seed):
    """The pass description rank `rank` of `world` hands to tghip_render_pass."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return capi.TgHipPassDesc(spp_begin, spp_end, seed & 0xFFFFFFFF, rank, world, 0)


def shard_skew(world):
    """include/tungsten_hip.h: tghip_shard_skew -- the smallest odd prime that does not divide the shard count."""
    for p in (3, 5, 7, 11, 13, 17):
        if world % p:
            return p
    return 1


def tile_owner(tx, ty, world):
    """include/tungsten_hip.h: tghip_tile_owner -- tile (tx, ty) of the 16x16 dicing belongs to shard (tx + ty*skew) % world."""
    return 0 if world <= 1 else (tx + ty*shard_skew(world)) % world


def owned_tiles(rank, world, width, height):
    """Row-major indices of the 16x16 tiles rank `rank` renders, in the order it renders them."""
    tiles_x, tiles_y = (width + 15)//16, (height + 15)//16
    return [tx + ty*tiles_x for ty in range(tiles_y) for tx in range(tiles_x) if tile_owner(tx, ty, world) == rank]


def reduce_framebuffer(fb_sum, fb_count, dst=0, group=None):
    """Sum-reduce radiance sums [H,W,3] float32 and sample counts [H,W] int32 to rank `dst` (in place there)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if fb_sum.is_cuda and dist.get_backend(group) != "nccl":
        # device buffers on a backend without device collectives (gloo: the CPU tests, and bench.py's rehearsal of N ranks on fewer GPUs):
        # stage through the host
        hs, hc = fb_sum.cpu(), fb_count.cpu()
        dist.reduce(hs, dst=dst, op=dist.ReduceOp.SUM, group=group)
        dist.reduce(hc, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank() == dst:                 # (`dst` of dist.reduce is a GLOBAL rank, whatever the group)
            fb_sum.copy_(hs)
            fb_count.copy_(hc)
        import torch
        torch.cuda.current_stream(fb_sum.device).synchronize()
        return
    dist.reduce(fb_sum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    dist.reduce(fb_count, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if fb_sum.is_cuda:
        # The collectives run on RCCL's own stream and the shim renders on its own HIP stream: the next
        # tghip_clear_framebuffer / pass must not start before the reduce has consumed (root: produced) the buffers.
        import torch
        torch.cuda.current_stream(fb_sum.device).synchronize()


def init_rank_comm(lib, ctx, rank, world, group=None):
    """The product's own exchange step for one process per GPU (include/tungsten_hip.h: tghip_comm_unique_id / tghip_comm_init_rank /
    tghip_reduce_framebuffer_rank -- ncclReduce between the ranks' contexts, no torch tensor involved).  torch.distributed only carries rank 0's
    128-byte RCCL id to the other ranks and agrees on the outcome: returns None when EVERY rank has its communicator, otherwise the reason
    (the same string on every rank) -- the caller then reduces with reduce_framebuffer() above, and says so.  Collective over `group`."""
    import ctypes as C
    import torch
    import torch.distributed as dist

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))
    n = capi.TGHIP_COMM_ID_BYTES
    buf = (C.c_ubyte*n)()
    probe = lib.tghip_comm_unique_id(buf, n)          # (every rank: does RCCL load here?  only rank 0's id is used)
    if not all_ok(probe == 0):
        return "tghip_comm_unique_id failed on a rank (librccl.so missing?)"
    box = [bytes(buf) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ident = (C.c_ubyte*n).from_buffer_copy(box[0])
    rc = lib.tghip_comm_init_rank(ctx, ident, n, world, rank)
    if not all_ok(rc == 0):
        return "tghip_comm_init_rank failed on a rank: %s" % (lib.tghip_last_error(ctx).decode() if rc != 0 else "(another rank)")
    return None


def merge_records(records, group=None):
    """SampleRecords (TGHIP_PASS_RECORDS) of a tile-sharded pass: every record is non-zero on exactly the rank that owns
    its tile, so an all-reduce(SUM) of the three fields hands every rank the complete, bit-exact set (x + 0).
    `records`: structured numpy array with sample_count (u32), mean, running_variance (f32); returned merged."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return records
    counts = torch.from_numpy(np.ascontiguousarray(records["sample_count"]).astype(np.int64))
    stats = torch.from_numpy(np.stack([records["mean"], records["running_variance"]]).astype(np.float32))
    if dist.get_backend(group) == "nccl":
        counts, stats = counts.cuda(), stats.cuda()
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    out = records.copy()
    out["sample_count"] = counts.cpu().numpy().astype(np.uint32)
    out["mean"], out["running_variance"] = stats.cpu().numpy()
    return out


def reduce_output_buffers(aux, dst=0, group=None):
    """Auxiliary output buffers (TGHIP_PASS_AUX) of a tile-sharded render: structured numpy array [H, W] of AUX_DTYPE
    (TgHipAuxPixel).  A pixel's record is non-zero only on the rank that owns its tile, so a sum-reduce to rank `dst`
    assembles the complete buffers exactly (x + 0).  Returns the merged array (meaningful on `dst`)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return aux
    floats = torch.from_numpy(np.concatenate([aux["a"], aux["b"], aux["variance"]], axis=-1).astype(np.float32))
    counts = torch.from_numpy(np.ascontiguousarray(aux["count"]).astype(np.int64))
    if dist.get_backend(group) == "nccl":
        floats, counts = floats.cuda(), counts.cuda()
    dist.reduce(floats, dst=dst, op=dist.ReduceOp.SUM, group=group)
    dist.reduce(counts, dst=dst, op=dist.ReduceOp.SUM, group=group)
    out = aux.copy()
    f = floats.cpu().numpy()
    out["a"], out["b"], out["variance"] = f[..., 0:11], f[..., 11:22], f[..., 22:33]
    out["count"] = counts.cpu().numpy().astype(np.uint32)
    return out


def render_loop(render_pass, download_records, width, height, spp, spp_step, seed, rank=0, world=1, adaptive=True, sobol=False,
                group=None, output_buffers=False):
    """The integrator's pass loop (PathTraceIntegrator::startRender / generateWork, PathTraceIntegrator.cpp:108-134,220-239)
    for one-process-per-GPU renders: every rank runs the same deterministic PassScheduler, renders its own tiles of each
    pass and the ranks exchange the SampleRecords of the pass (merge_records) before the next generateWork.

    render_pass(TgHipPassDesc) renders one pass on this rank (tghip_render_pass + tghip_wait, or a stand-in);
    download_records() returns this rank's device records (tghip_download_records layout).  Returns the scheduler."""
    import ctypes as C
    import numpy as np
    from . import PassScheduler
    sch = PassScheduler(width, height, seed)
    tile_seeds = np.ascontiguousarray(sch.tile_seeds)
    cur = 0
    while cur < spp:
        nxt = min(cur + spp_step, spp)
        if sch.generate_work(cur, nxt, adaptive):
            rec = sch.records
            p = shard_pass(rank, world, cur, nxt, seed)
            p.flags = ((capi.TGHIP_PASS_SOBOL if sobol else 0) | (capi.TGHIP_PASS_RECORDS if adaptive else 0)
                       | (capi.TGHIP_PASS_AUX if output_buffers else 0))     # renderer.output_buffers: reduce_output_buffers afterwards
            index = np.ascontiguousarray(rec["sample_index"])
            count = np.ascontiguousarray(rec["next_sample_count"])
            if sobol:
                p.tile_seeds = tile_seeds.ctypes.data_as(C.POINTER(C.c_uint32))
            if adaptive:
                p.record_index = index.ctypes.data_as(C.POINTER(C.c_uint32))
                p.record_count = count.ctypes.data_as(C.POINTER(C.c_uint32))
            render_pass(p)
            if adaptive:
                merged = merge_records(download_records(), group=group)
                for f in ("sample_count", "mean", "running_variance"):
                    rec[f] = merged[f]
        cur = nxt
    return sch
