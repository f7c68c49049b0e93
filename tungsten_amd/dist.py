"""Multi-GPU decomposition of a render pass: one process per GPU, 16x16 tiles dealt round-robin
to ranks (the reference's own dicing, integrators/path_tracer/PathTraceIntegrator.cpp:27-42), and ONE exchange
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


def owned_tiles(rank, world, width, height):
    """Row-major indices of the 16x16 tiles rank `rank` renders."""
    tiles = ((width + 15)//16)*((height + 15)//16)
    return range(rank, tiles, world)


def reduce_framebuffer(fb_sum, fb_count, dst=0, group=None):
    """Sum-reduce radiance sums [H,W,3] float32 and sample counts [H,W] int32 to rank `dst` (in place there)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.reduce(fb_sum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    dist.reduce(fb_count, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if fb_sum.is_cuda:
        # The collectives run on RCCL's own stream and the shim renders on its own HIP stream: the next
        # tghip_clear_framebuffer / pass must not start before the reduce has consumed (root: produced) the buffers.
        import torch
        torch.cuda.current_stream(fb_sum.device).synchronize()
