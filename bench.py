#!/usr/bin/env python3
"""bench.py -- throughput of the path_tracer_hip hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

A "step" is one complete render of the workload through the C-ABI: tghip_clear_framebuffer +
tghip_render_pass(spp 0..S) + tghip_wait on every rank's tile shard (16x16 tiles dealt round-robin,
SURVEY.md 8e) and, for N > 1, the RCCL sum-reduce of the float framebuffer to rank 0.  The scene is
flattened and uploaded before the timed region (inputs resident in HBM); the framebuffer stays in HBM.

Workload at every N: the configuration BASELINE.json's metric is quoted on -- materialtest.json (the reference's shipped
scene: three meshes, 80 768 triangles, smooth_coat over rough_conductor, HDRI environment + MIS) at 1280x720, 256 spp,
uniform sampler, adaptive sampling off (fixed total work => "scaling": "strong").  At N = 1 the same line carries, under
"extra", BASELINE configs[1] (Cornell box 1280x720 at 256 spp, a flat-list scene without BVH traversal) and materialtest as the reference ships it (Sobol + adaptive, 64 spp in 16-spp passes); `--scene cornell`
makes that the headline workload instead.  Without the materialtest assets (assets/, copied from the reference's
data directory by __graft_entry__.build()) the default run FAILS: there is no silent fallback to another workload.

Prints ONE JSON line (rank 0).  `roofline` describes the traversal kernel BASELINE.json's metric names (the traversal class with the
largest accumulated time; flat-list scenes: their one fused kernel): achieved = algorithmic bytes per launch / average launch duration,
both from the timed region (HIP events recorded by the shim on the stream the kernels run on; byte model in DESIGN.md section 5).
`roofline.dominant_class` holds the same for the class with the largest accumulated time of any kind (k_shade on the headline workload),
`roofline.exclusive` the figures of every class with the chip to itself (one part on one stream), `roofline.valu` the loop's VALU issue
line, plain and priced per instruction category, with the enabled lanes per issued instruction and the walks' turn tallies.  `cpu_baseline`
times the reference itself (oracle/_ref/tungsten, kind "reference") or, when that binary is absent, the
oracle port, on a bounded sample of the same workload on this box's host cores.
"""
try:
    # FIRST, before anything that loads libtungsten_hip.so: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  Loaded first, it is the one
    # copy in the process (the library's DT_NEEDED libamdhip64.so.7 resolves to it by SONAME); loaded second, the process ends up with two HSA
    # runtimes and the second one finds no device (round 6 lost two GPU sessions to `from tungsten_amd import workloads` ahead of `import torch`).
    import torch  # noqa: F401
except ImportError:
    torch = None
import argparse
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MAX_CLOCK_HZ = 2.4e9       # MI355X_MICROARCH.md: max engine clock


class ClockPoller(object):
    """The shader clock while the timed region runs: `rocm-smi --showclocks` called back to back on a host thread (about six readings a second), the
    sclk of the busiest visible GPU per reading.  The VALU ceiling below is priced at the MAXIMUM clock; this says how far under it the chip ran."""
    def __init__(self):
        import threading
        self.readings, self.stop_flag = [], False
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self.stop_flag:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=10).stdout
            except Exception:
                return
            mhz = [int(m) for m in re.findall(r"sclk clock level[^(]*\((\d+)Mhz\)", out, re.I)]
            if mhz:
                self.readings.append(max(mhz))

    def start(self):
        self.thread.start()
        return self

    def stop(self):
        self.stop_flag = True
        self.thread.join(timeout=15)
        r = sorted(v for v in self.readings if v >= 1000)           # (readings taken while the clock ramps up from idle are not the loop's)
        if not r:
            return None
        return {"mhz_median": r[len(r)//2], "mhz_min": r[0], "mhz_max": r[-1], "readings": len(r), "frac_of_max_clock": round(r[len(r)//2]*1e6/MAX_CLOCK_HZ, 4),
                "source": "rocm-smi --showclocks polled on a host thread during the timed region (sclk of the busiest visible GPU)"}
VALU_CYCLES_PER_WAVE64 = 2.0   # MI355X_MICROARCH.md "Per-instruction cycle constants": v_fma_f32 (wave64) = 2 cycles (the CDNA4 SIMD is 32 lanes wide)
# What the OTHER wave64 VALU instructions cost per SIMD, measured against v_fma_f32 = 2 with tools/ubench_valu.hip (profiles/r5_ubench_valu.txt):
# f32 add / mul / fma, and / or / add_u32 / mov run at 1.8-2.0; min / max (also the 3-operand forms), shifts, bfe, every convert, compares,
# integer multiplies, bit counts, v_perm, v_fma_mix_f32 at 3.1-3.4; packed f32 3.6-3.7 (no gain over two scalar ones); f64 fma 3.8; rcp / sqrt 6.2.
# The counters below split SQ_INSTS_VALU by category; INT32 and the remainder ("other": moves, f32 compares and selects, min / max, lane ops)
# mix both price classes, so they carry a low and a high price.
VALU_PRICE = {"SQ_INSTS_VALU_ADD_F32": (2.0, 2.0), "SQ_INSTS_VALU_MUL_F32": (2.0, 2.0), "SQ_INSTS_VALU_FMA_F32": (2.0, 2.0),
              "SQ_INSTS_VALU_TRANS_F32": (6.2, 6.2), "SQ_INSTS_VALU_CVT": (3.2, 3.2), "SQ_INSTS_VALU_INT32": (1.8, 3.2),
              "SQ_INSTS_VALU_INT64": (3.2, 3.8), "SQ_INSTS_VALU_FMA_F64": (3.8, 3.8), "SQ_INSTS_VALU_ADD_F64": (3.8, 3.8),
              "SQ_INSTS_VALU_MUL_F64": (3.8, 3.8), "other": (1.85, 3.2)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="materialtest", choices=["cornell", "materialtest", "mesh1m", "instances10k"])
    ap.add_argument("--material", default="shipped", choices=["shipped", "dielectric", "rough_dielectric"],
                    help="materialtest only: the \"Material\" bsdf (BASELINE configs[2] names rough-conductor -- the shipped one -- and dielectric)")
    ap.add_argument("--res", default="1280x720")
    ap.add_argument("--spp", type=int, default=0, help="default: 256 (materialtest, cornell) / 32 (mesh1m, instances10k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary Cornell-box run")
    ap.add_argument("--no-clock", dest="clock", action="store_false", help="do not poll rocm-smi for the shader clock during the timed region (N = 1)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record per-launch HIP events")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="target CPU time of the cpu_baseline sample (three runs of two builds together)")
    ap.add_argument("--no-traffic", dest="traffic", action="store_false",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--no-exclusive", dest="exclusive", action="store_false",
                    help="skip the two extra steps with the pool as one part on one stream (roofline.exclusive: every launch with the chip to itself)")
    ap.add_argument("--opt", action="append", default=[], help="shim option key=value (tghip_set_option)")
    ap.add_argument("--count-spp", type=int, default=0,
                    help="spp of the untimed counting pass that feeds the byte model (default: the workload's own spp, i.e. exact counts; a lower "
                         "value scales the counts by spp/count_spp -- for the multi-minute configurations)")
    ap.add_argument("--reduce", default="product", choices=["product", "torch"],
                    help="N > 1: the exchange step -- the product's own tghip_reduce_framebuffer_rank (ncclReduce between the ranks' contexts; falls back to "
                         "torch.distributed.reduce, and says so, if a rank cannot build its communicator) or torch.distributed.reduce on the bound tensors")
    ap.add_argument("--in-process", action="store_true",
                    help="N > 1 in ONE process (no torch.distributed.run): the host integrator drives N contexts from N host threads (\"devices\": N) and merges "
                         "with tghip_reduce_framebuffers; a step = one complete render through tgh_renderer_render, download of the merged image included")
    ap.add_argument("--emulate-shards", type=int, default=0,
                    help="development aid: on ONE GPU render and time EVERY one of N tile shards in turn (what the ranks of an N-GPU run do; "
                         "reports max / mean / min over shards and the fixed cost of the framebuffer reduce)")
    return ap.parse_args()


# ---- algorithmic bytes -------------------------------------------
NODE_B, REC_B = 64, 48          # TgHipBvhNode, TgHipPrimRec
WIDE_NODE_B = 80                # TgHipWideNode
RAY_B, HIT_B = 32, 16           # (o,tmin,d,tmax), (t,u,v,rec)
STATE_R = RAY_B + 16 + 16 + 16  # ray, throughput+flags, radiance, rng+pixel+item read per shaded vertex
STATE_W = 16 + 16               # throughput+flags, radiance written back per shaded vertex


def kernel_bytes(c, flat, fused, node_b=64, node_b_shadow=None, fold_finish=False):
    """Algorithmic bytes moved by each kernel class over everything the counters cover.
    flat: the scene is a flat record list whose loads are wave-uniform (one fetch per 64 rays);
    fused: (flat scenes) intersection and shadow tests run inside k_shade, no hit / shadow records exist."""
    nodes_sh, prims_sh = c["nodes_visited_shadow"], c["prims_tested_shadow"]
    nodes_cl, prims_cl = c["nodes_visited"] - nodes_sh, c["prims_tested"] - prims_sh
    rec_b = REC_B/64.0 if flat else REC_B
    paths = c["closest_rays"]            # path vertices == extension rays
    alive = max(c["closest_rays"] - c["samples"], 0)
    regen = c["samples"]*(RAY_B + 16 + 16 + 16 + 16 + 16)   # a finished sample rewrites the whole slot record
    if fused:
        return {"k_shade": paths*(STATE_R + STATE_W) + alive*(RAY_B + 8) + regen + rec_b*c["prims_tested"]}
    return {
        # ray in + hit out + BVH nodes and primitive records actually visited
        # (fold_finish: the launch also finalises and regenerates the slots of the samples that ended at the previous vertex -- k_finish's
        # work rides in front of the walk: the sample's radiance and flags read, the slot record rewritten)
        "k_trace_closest": paths*(RAY_B + HIT_B) + node_b*nodes_cl + rec_b*prims_cl + (c["samples"]*(16 + 16) + regen if fold_finish else 0),
        # shadow origin, throughput/pending, radiance read+write per slot; direction+contribution per ray
        "k_trace_shadow": c["shadow_slots"]*(16 + 16 + 16 + 32) + c["shadow_rays"]*32 + (node_b_shadow or node_b)*nodes_sh + rec_b*prims_sh,
        # path state read once per vertex and written back (+ ray and rng when the path continues), plus the
        # shadow records a vertex emits
        "k_shade": paths*(STATE_R + HIT_B + STATE_W) + alive*(RAY_B + 8) + c["shadow_slots"]*(16 + 64 + 16 + 16),
    }


def as_shipped(tmp, tg_mod, repeats=3):
    from tungsten_amd import workloads as scenes
    path = scenes.materialtest(tmp, name="as_shipped.json", resolution=(1280, 720), spp=64, spp_step=16,
                               renderer={"adaptive_sampling": True, "stratified_sampler": True})
    best = None
    for _ in range(repeats):
        r = tg_mod.Renderer(path)
        secs = r.render()
        c = r.counters()
        mean, _, count = r.image()
        r.close()
        res = {"value": round(c.samples/secs*1e-6, 2), "unit": "Msamples/s", "seconds": round(secs, 4), "samples": int(c.samples), "passes": 4,
               "count_min": int(count.min()), "count_max": int(count.max()), "sampler": "sobol", "adaptive_sampling": True,
               "image_mean": [round(float(v), 6) for v in mean.mean(axis=(0, 1))]}
        if best is None or res["seconds"] < best["seconds"]:
            best = res
    return best


def walk_summary(t):
    """tghip_get_walk_stats of one decoupled walk (the counting variants' tallies, include/tungsten_hip.h) -> what a ray costs in loop turns and
    how many of a wave's 64 lanes have work in each section of a turn.  A section's instructions are issued for the whole wave whenever ANY lane
    needs it: `lanes_per_run` / 64 is the section's lane utilisation, `runs_per_turn` how often a turn pays for it."""
    if len(t) >= 48 and t[24]:            # instanced scenes: runs / lanes of the turn's sections
        names = ("turn", "node", "node_level1", "leaf", "instance_leaf", "set_entry", "triangle_leaf", "pop", "master_done", "instance_tree_pop", "publish", "refill")
        if t[22]:                         # k_trace_closest_instw (masters through the wide BVH, phases voted on); else k_trace_closest_inst
            names = ("turn", "phase_master", "master_record_test", "master_node_visit", "master_done", "phase_tree", "tree_node_step", "tree_leaf_section",
                     "instance_leaf", "master_entered", "pop", "refill")
        return {"kernel": "k_trace_closest_instw" if t[22] else "k_trace_closest_inst", "wide_master_nodes": t[23],
                "wave_launches": t[4], "turns_per_wave_launch": round(t[24]/max(t[4], 1), 2),
                "sections": {n: {"runs_per_turn": round(t[24 + 2*k]/float(t[24]), 4), "lanes_per_run": round(t[25 + 2*k]/max(t[24 + 2*k], 1), 2)} for k, n in enumerate(names)}}
    turns = t[5] + t[6]
    rays = max(t[21] + t[10], 1)          # walks started + walks resumed
    def sec(runs, lanes):
        return {"runs_per_turn": round(runs/max(turns, 1), 4), "lanes_per_run": round(lanes/max(runs, 1), 2), "utilisation": round(lanes/max(runs, 1)/64.0, 4)}
    return {"wave_launches": t[4], "turns_per_wave_launch": round(turns/max(t[4], 1), 2), "turns_after_queue_dry": round(t[6]/max(turns, 1), 4),
            "busy_lanes_per_turn": round((t[7] + t[8])/max(turns, 1), 2), "rays": t[21], "walks_resumed": t[10],
            "turns_per_ray": round((t[7] + t[8])/rays, 3),
            "record_test": sec(t[12], t[13]), "node_visit": sec(t[14], t[15]), "refill": sec(t[16], t[17]), "publish": sec(t[18], t[19]),
            "records_per_ray": round(t[13]/rays, 3), "nodes_per_ray": round(t[15]/rays, 3), "hits_accepted_per_ray": round(t[20]/rays, 3),
            "us_per_wave_launch": {"expand": round(t[0]*0.01/max(t[4], 1), 2), "loop_with_queue": round(t[1]*0.01/max(t[4], 1), 2),
                                   "loop_dry": round(t[2]*0.01/max(t[4], 1), 2), "wait_writeback": round(t[3]*0.01/max(t[4], 1), 2)}}


def ordered_roofline(r):
    """The same dictionary with the judged fields and the exclusive-launch summary in front of the bulky diagnostics (a reader that keeps only the head of the
    JSON line -- the driver's record did in round 5 -- still sees them)."""
    if not isinstance(r, dict):
        return r
    head = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_us", "launches", "concurrent_parts", "exclusive", "loop",
            "timing", "schema", "note")
    out = {k: r[k] for k in head if k in r}
    out.update({k: v for k, v in r.items() if k not in out})
    return out


def counters_dict(c):
    return {k: getattr(c, k) for k, _ in c._fields_}


class Bench(object):
    def __init__(self, a):
        import torch
        import tungsten_amd as tg
        self.a, self.torch, self.tg = a, torch, tg
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus and not (a.in_process and self.world == 1):
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)" % (a.gpus, self.world))
        if not torch.cuda.is_available() or tg.device_count() < 1:
            raise SystemExit("bench.py: no HIP device visible -- the path tracer has no CPU fallback")
        # TG_BENCH_SHARE_DEVICE=1 (a rehearsal of the N-rank path on a box with fewer GPUs than ranks, profiles/README.md): every rank renders
        # its shard on device LOCAL_RANK % visible devices and the exchange step goes through gloo -- RCCL wants one rank per device
        self.shared = os.environ.get("TG_BENCH_SHARE_DEVICE", "") not in ("", "0")
        if self.shared:
            self.local %= torch.cuda.device_count()
        torch.cuda.set_device(self.local)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.shared:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local))
            self.dist = dist
        self.tmp = tempfile.mkdtemp(prefix="tg_bench_")

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
        shutil.rmtree(self.tmp, ignore_errors=True)

    def fence(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def make_scene(self, scene, w, h, spp):
        """Writes the workload's scene description (tungsten_amd/workloads.py) into the run's scratch directory: (path, description)."""
        from tungsten_amd import workloads as scenes
        a = self.a
        if scene == "materialtest":
            if not scenes.have_materialtest():
                raise SystemExit("bench.py: materialtest assets missing (assets/; run __graft_entry__.build() where the reference is mounted)")
            edit = None
            if a.material == "dielectric":
                edit = scenes._mt_material({"type": "dielectric", "ior": 1.5, "albedo": 1})
            elif a.material == "rough_dielectric":
                edit = scenes._mt_material({"type": "rough_dielectric", "ior": 1.5, "distribution": "ggx", "roughness": 0.1, "albedo": 1})
            path = scenes.materialtest(self.tmp, resolution=(w, h), spp=spp, edit=edit)
            workload = "materialtest.json (3 meshes 80768 tris + quad, smooth_coat/%s/lambert, envmap MIS) %dx%d @ %d spp" % (
                "rough_conductor" if a.material == "shipped" else a.material, w, h, spp)
        elif scene == "mesh1m":
            path = scenes.mesh1m(self.tmp, resolution=(w, h), spp=spp)
            workload = ("BASELINE configs[3] on one GPU: procedurally generated 998 000-triangle mesh (fixed seed 1) + floor quad, rough_conductor, "
                        "HDRI environment + MIS, %dx%d @ %d spp" % (w, h, spp))
        elif scene == "instances10k":
            path = scenes.instances10k(self.tmp, resolution=(w, h), spp=spp)
            workload = ("BASELINE configs[4] scaled to one GPU: 10 000 rigid instances of a 19 800-triangle mesh (4 masters: lambert / rough_conductor / "
                        "dielectric / plastic), HDRI environment + MIS, %dx%d @ %d spp" % (w, h, spp))
        else:
            path = scenes.cornell(self.tmp, resolution=(w, h), spp=spp)
            workload = "BASELINE configs[1]: cornell-box (5 quads + 2 cubes + quad light, Lambert) %dx%d @ %d spp" % (w, h, spp)
        return path, workload

    def run(self, scene, w, h, spp, steps, warmup, cpu):
        """Times `steps` renders of `scene`; returns the result dict on rank 0 (None elsewhere)."""
        import numpy as np
        from tungsten_amd import workloads as scenes
        from tungsten_amd import dist as tgdist
        a, tg, torch, lib = self.a, self.tg, self.torch, self.tg.lib
        path, workload = self.make_scene(scene, w, h, spp)

        t0 = time.time()
        flat = tg.FlattenedScene(path)
        t_flatten = time.time() - t0
        # (tools/sweep.py renders every option set on ONE context: HIP deals a context's streams to the hardware queues in creation
        # order, and a second context of the same process gets a mapping that serialises parts of the loop -- 675 against 840 Msamples/s)
        shared = getattr(self, "shared_ctx", None)
        ctx = shared or lib.tghip_create(self.local)
        if not ctx:
            raise SystemExit("tghip_create: " + lib.tghip_last_error(None).decode())

        def check(rc, what):
            if rc != 0:
                raise SystemExit("%s failed (%d): %s" % (what, rc, lib.tghip_last_error(ctx).decode()))
        for kv in a.opt:                          # (before the upload: some options shape the uploaded scene)
            k, v = kv.split("=")
            check(lib.tghip_set_option(ctx, k.encode(), int(v)), "tghip_set_option")
        t0 = time.time()
        check(lib.tghip_upload_scene(ctx, flat.desc), "tghip_upload_scene")
        t_upload = time.time() - t0

        # framebuffer lives in torch tensors so that RCCL (torch.distributed) can reduce it in place
        fb_sum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        fb_cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        check(lib.tghip_bind_framebuffer(ctx, fb_sum.data_ptr(), fb_cnt.data_ptr()), "tghip_bind_framebuffer")
        pass_desc = tgdist.shard_pass(self.rank, self.world, 0, spp, tg.DEFAULT_SEED)
        # N > 1: the exchange step is the product's own (one ncclComm_t per context, ncclReduce of the framebuffers into a scratch image on rank 0's
        # device); torch.distributed carries the RCCL id, the barriers and the timing all-reduce.  The rehearsal with ranks sharing a device stays
        # on gloo: RCCL wants one rank per device.
        product_reduce, reduce_note = False, None
        if self.dist is not None:
            if self.shared:
                reduce_note = "gloo (rehearsal: ranks share a device)"
            elif a.reduce == "product":
                why = tgdist.init_rank_comm(lib, ctx, self.rank, self.world)
                product_reduce = why is None
                reduce_note = ("tghip_reduce_framebuffer_rank (the product's ncclReduce between the ranks' contexts)" if product_reduce
                               else "torch.distributed.reduce -- FALLBACK, the product's communicator could not be built: %s" % why)
            else:
                reduce_note = "torch.distributed.reduce (--reduce torch)"
        emulated = None
        if a.emulate_shards > 1 and self.world == 1:
            # what an N-GPU run does, on ONE GPU: every shard 0..N-1 is rendered and timed in turn (a strong-scaling run is as
            # slow as its slowest rank, so the line's value is priced by the MAX over shards), plus the exchange step -- the
            # RCCL reduce behind the C-ABI with one rank, i.e. its fixed cost: the device-side reduce + the PCIe download
            def time_shard(r):
                pd = tgdist.shard_pass(r, a.emulate_shards, 0, spp, tg.DEFAULT_SEED)
                for _ in range(max(warmup, 1)):
                    check(lib.tghip_clear_framebuffer(ctx), "clear"); check(lib.tghip_render_pass(ctx, C.byref(pd)), "render"); check(lib.tghip_wait(ctx), "wait")
                self.fence()
                t0 = time.perf_counter()
                for _ in range(steps):
                    check(lib.tghip_clear_framebuffer(ctx), "clear"); check(lib.tghip_render_pass(ctx, C.byref(pd)), "render"); check(lib.tghip_wait(ctx), "wait")
                self.fence()
                return (time.perf_counter() - t0)/steps*1e3
            per_shard = [time_shard(r) for r in range(a.emulate_shards)]
            import numpy as _np
            hs, hc = _np.empty((h, w, 3), _np.float32), _np.empty((h, w), _np.uint32)
            ctxs = (C.c_void_p*1)(ctx)
            red = []
            for _ in range(5):
                t0 = time.perf_counter()
                rc = lib.tghip_reduce_framebuffers(ctxs, 1, 0, hs.ctypes.data, hc.ctypes.data, w*h)
                red.append((time.perf_counter() - t0)*1e3)
                if rc != 0:
                    red = None
                    break
            emulated = {"shards": a.emulate_shards, "ms_per_step_by_shard": [round(t, 3) for t in per_shard],
                        "max_ms": round(max(per_shard), 3), "mean_ms": round(sum(per_shard)/len(per_shard), 3), "min_ms": round(min(per_shard), 3),
                        "reduce_n1_ms": round(sorted(red)[len(red)//2], 3) if red else None,
                        "note": "value and ms_per_step below are shard 0's; the estimate of an N-GPU step is max_ms + the reduce"}
            pass_desc = tgdist.shard_pass(0, a.emulate_shards, 0, spp, tg.DEFAULT_SEED)

        split = [0.0, 0.0]                       # this rank's seconds in its shard's render / in the exchange step, over the timed region

        def step():
            t_a = time.perf_counter()
            check(lib.tghip_clear_framebuffer(ctx), "tghip_clear_framebuffer")
            check(lib.tghip_render_pass(ctx, C.byref(pass_desc)), "tghip_render_pass")
            check(lib.tghip_wait(ctx), "tghip_wait")
            t_b = time.perf_counter()
            # the exchange step: float framebuffer sum-reduce over xGMI (tile ownership is disjoint -> exact)
            if product_reduce:
                check(lib.tghip_reduce_framebuffer_rank(ctx, 0, None, None, w*h), "tghip_reduce_framebuffer_rank")   # (the merged image stays in HBM)
            else:
                tgdist.reduce_framebuffer(fb_sum, fb_cnt, dst=0)
            split[0] += t_b - t_a
            split[1] += time.perf_counter() - t_b

        for _ in range(warmup):
            step()
        check(lib.tghip_reset_counters(ctx), "tghip_reset_counters")
        check(lib.tghip_set_option(ctx, b"time_kernels", 0 if a.no_kernel_timing else 1), "tghip_set_option")
        self.fence()
        split[0] = split[1] = 0.0
        # (never in the counter passes' child runs -- they pass --no-clock: a rocm-smi process started under rocprofv3's preloaded tool library takes the run down)
        poller = ClockPoller().start() if (self.world == 1 and a.clock and a.steps > 0 and shutil.which("rocm-smi") and "ROCPROFILER" not in " ".join(os.environ)
                                           and "rocprof" not in os.environ.get("LD_PRELOAD", "")) else None
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        elapsed = time.perf_counter() - t0
        sustained_clock = poller.stop() if poller else None
        per_rank = None
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if self.shared else "cuda")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            # every rank's own clock, so that a SCALE line explains itself: the shard's render, the exchange step (which includes waiting for
            # the slowest rank: the reduce is a collective), and the whole timed region
            mine = torch.tensor([split[0], split[1], elapsed], dtype=torch.float64, device="cpu" if self.shared else "cuda")
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(every, mine)
            per_rank = {"render_ms_per_step": [round(float(e[0])/steps*1e3, 3) for e in every],
                        "reduce_ms_per_step": [round(float(e[1])/steps*1e3, 3) for e in every],
                        "total_ms_per_step": [round(float(e[2])/steps*1e3, 3) for e in every],
                        "note": "each rank's host clock over the timed region; reduce = the framebuffer sum-reduce to rank 0, a collective: it also "
                                "holds the time a rank waits for the slowest one"}
            elapsed = float(t.item())
        timed = tg.TgHipCounters()
        lib.tghip_get_counters(ctx, C.byref(timed))
        timed = counters_dict(timed)
        check(lib.tghip_set_option(ctx, b"time_kernels", 0), "tghip_set_option")

        merged = None
        if product_reduce:
            # (outside the timed region: the merged image of the last step, downloaded from rank 0's scratch image -- a collective, every rank calls)
            hs = np.empty((h, w, 3), np.float32) if self.rank == 0 else None
            hc = np.empty((h, w), np.uint32) if self.rank == 0 else None
            check(lib.tghip_reduce_framebuffer_rank(ctx, 0, hs.ctypes.data if hs is not None else None, hc.ctypes.data if hc is not None else None, w*h),
                  "tghip_reduce_framebuffer_rank")
            merged = (hs, hc)
        out = None
        if self.rank == 0:
            # sanity of the result of the last timed step (rank 0 holds the reduced image)
            if merged:
                cnt = merged[1].astype(np.int64)
                img = merged[0]/np.maximum(cnt, 1)[..., None]
            else:
                cnt = fb_cnt.cpu().numpy()
                img = (fb_sum/fb_cnt.clamp(min=1).unsqueeze(-1)).cpu().numpy()
            ok = bool(((cnt == spp).all() or a.emulate_shards > 1) and np.isfinite(img).all())
            value = float(w)*h*spp*steps/elapsed*1e-6

            # one untimed counting step: exact node / record visit counts of the same (deterministic) render
            check(lib.tghip_set_option(ctx, b"count_traversal", 1), "tghip_set_option")
            check(lib.tghip_reset_counters(ctx), "tghip_reset_counters")
            check(lib.tghip_clear_framebuffer(ctx), "clear")
            count_spp = min(spp, a.count_spp) if a.count_spp > 0 else spp
            count_desc = pass_desc if count_spp == spp else tgdist.shard_pass(self.rank, self.world, 0, count_spp, tg.DEFAULT_SEED)
            check(lib.tghip_render_pass(ctx, C.byref(count_desc)), "render")
            check(lib.tghip_wait(ctx), "wait")
            cc = tg.TgHipCounters()
            lib.tghip_get_counters(ctx, C.byref(cc))
            cc = counters_dict(cc)
            walk_stats = {}
            for wi, wname in ((0, "closest_hit"), (1, "shadow")):
                buf = (C.c_uint64*48)()
                nw = lib.tghip_get_walk_stats(ctx, wi, buf, 48)
                if nw >= 22 and buf[4]:
                    walk_stats[wname] = walk_summary([int(v) for v in buf[:nw]])
            if count_spp != spp:                 # (the first count_spp samples of every pixel stand for all of them)
                cc = {k: (v*spp//count_spp if isinstance(v, int) else v*spp/count_spp) for k, v in cc.items()}
            check(lib.tghip_set_option(ctx, b"count_traversal", 0), "tghip_set_option")
            is_flat = int(flat.info.num_recs) <= 16
            fused = is_flat and cc["shadow_slots"] == 0 and cc["shadow_rays"] > 0
            # BVH scenes walk the 8-wide BVH (80-byte nodes) unless the run switched it off
            wide = int(flat.desc.contents.num_wide_nodes) > 0 and "wide_bvh=0" not in a.opt
            # (instanced scenes: shadow rays on the wide tree, closest-hit rays on the two-level BVH2 -- the shim's measured default)
            inst = int(flat.desc.contents.num_instances) > 0
            # single-level BVH scenes on the decoupled wide walk: k_finish is folded into the closest-hit launch (the shim's default)
            fold = wide and not inst and not is_flat and "fold_finish=0" not in a.opt and "decouple=0" not in a.opt
            per_step_bytes = kernel_bytes(cc, is_flat, fused, WIDE_NODE_B if wide and not inst else NODE_B, WIDE_NODE_B if wide else NODE_B, fold)
            if inst and walk_stats.get("closest_hit", {}).get("wide_master_nodes"):      # (k_trace_closest_instw: the masters' nodes are 80-byte wide nodes)
                per_step_bytes["k_trace_closest"] += (WIDE_NODE_B - NODE_B)*walk_stats["closest_hit"]["wide_master_nodes"]*(spp//count_spp if count_spp != spp else 1)

            kernels = {}
            ms = {"k_trace_closest": timed["ms_trace_closest"], "k_shade": timed["ms_shade"], "k_trace_shadow": timed["ms_trace_shadow"]}
            launches = {"k_trace_closest": timed["launches_trace_closest"], "k_shade": timed["launches_shade"],
                        "k_trace_shadow": timed["launches_trace_shadow"]}
            for k in per_step_bytes:
                if launches[k] and ms[k] > 0:
                    bytes_per_launch = per_step_bytes[k]*steps/launches[k]
                    avg_s = ms[k]*1e-3/launches[k]
                    kernels[k] = {"ms_total": round(ms[k], 3), "launches": int(launches[k]), "avg_us": round(avg_s*1e6, 2),
                                  "bytes_per_launch": round(bytes_per_launch), "gbs": round(bytes_per_launch/avg_s*1e-9, 1)}
            roofline = None
            if kernels:
                # The headline fields describe the TRAVERSAL kernel class with the largest accumulated time over the timed region -- the kernel
                # BASELINE.json's metric asks the achieved GB/s of (rounds 1-4 and 6; round 5's lines had the class with the largest accumulated
                # time of ANY kind there, k_shade on the metric's workload: that one is under `dominant_class` now).  Flat-list scenes have one
                # fused kernel, which is both.
                trav = [k for k in kernels if k.startswith("k_trace")]
                dom = max(list(kernels), key=lambda k: kernels[k]["ms_total"])
                dom_trav = max(trav, key=lambda k: kernels[k]["ms_total"]) if trav else None
                head = dom_trav or dom
                kd = kernels[head]
                roofline = {"bound": "hbm", "kernel": head + (" (trace + shade + shadow fused, flat-list scene)" if fused else ""),
                            "achieved": kd["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kd["gbs"]/HBM_PEAK_GBS, 4),
                            "traffic": None, "bytes_per_launch": kd["bytes_per_launch"], "avg_launch_us": kd["avg_us"],
                            "launches": kd["launches"],
                            "timing": "HIP events per launch on the shim's stream, over the timed region",
                            "schema": "r6: top-level fields = the traversal kernel (as rounds 1-4); the class with the largest accumulated time is `dominant_class`"}
                # The shim runs the pool as P parts on P streams (tungsten_hip.hip: runBatch); launches of different parts share the CUs, so a
                # launch has a fraction of the chip for its duration.  `loop` prices the whole loop instead: the algorithmic bytes of all
                # its kernels over the wall time of the timed region.
                parts = max(1, int(round(kd["launches"]/max(timed["iterations"], 1))))
                if parts > 1:
                    roofline["concurrent_parts"] = parts
                loop_bytes = sum(per_step_bytes[k] for k in kernels)*steps
                roofline["loop"] = {"achieved": round(loop_bytes/elapsed*1e-9, 1), "unit": "GB/s", "frac": round(loop_bytes/elapsed*1e-9/HBM_PEAK_GBS, 4),
                                    "bytes_per_iteration": round(loop_bytes/max(timed["iterations"], 1)),
                                    "us_per_iteration": round(elapsed/max(timed["iterations"], 1)*1e6, 1)}
                if fused:
                    roofline["note"] = ("instruction-bound, not HBM-bound (profiles/r1/sq_counters.json, profiles/r5_sq_counters.json): exact fp32 division/sqrt/sin/cos "
                                        "and -ffp-contract=off for parity with the CPU reference (DESIGN.md sections 5, 7)")
                if dom != head:
                    kt = kernels[dom]
                    roofline["dominant_class"] = {"kernel": dom, "achieved": kt["gbs"], "frac": round(kt["gbs"]/HBM_PEAK_GBS, 4),
                                                  "bytes_per_launch": kt["bytes_per_launch"], "avg_launch_us": kt["avg_us"], "launches": kt["launches"],
                                                  "traffic": None}
                if parts > 1:
                    roofline["note"] = ("shared-chip figure: the launches of %d parts of the pool overlap on the CUs, so a launch's duration is residency, not "
                                        "cost; `exclusive` holds the same kernels with the chip to themselves" % parts)
                # every kernel class of the loop priced the same way (the headline fields above are the dominant traversal kernel's)
                roofline["per_kernel"] = {k: {"achieved": kernels[k]["gbs"], "frac": round(kernels[k]["gbs"]/HBM_PEAK_GBS, 4),
                                              "bytes_per_launch": kernels[k]["bytes_per_launch"], "avg_launch_us": kernels[k]["avg_us"],
                                              "traffic": None} for k in sorted(kernels)}
                if a.traffic and self.world == 1:
                    traffic, source = measure_traffic(a, scene, w, h, spp, self.tmp)
                    roofline["traffic_source"] = source
                    for k in roofline["per_kernel"]:
                        if k in traffic:
                            roofline["per_kernel"][k]["traffic"] = traffic[k]["bytes_per_launch"]
                            roofline["per_kernel"][k]["traffic_launches"] = traffic[k]["launches"]
                    if "dominant_class" in roofline and dom in traffic:
                        roofline["dominant_class"]["traffic"] = traffic[dom]["bytes_per_launch"]
                    if head in traffic:
                        roofline["traffic"] = traffic[head]["bytes_per_launch"]
                    elif traffic:
                        # (a counter file without the headline kernel is a bug of the name matching, not a measurement)
                        raise SystemExit("bench.py: no %s dispatches in the counter file; classes seen: %s" % (head, sorted(traffic)))
                if a.traffic and self.world == 1 and not fused:
                    # What the loop of a BVH scene IS bound by: none of its kernels moves bytes at a rate worth pricing
                    # against HBM -- the trees are cache-resident -- so two more ceilings are reported next to the HBM line.
                    #  valu:      wave-wide VALU instructions of ALL kernels of the loop (one more counter pass, SQ_INSTS_VALU, of the same
                    #             workload) over the wall clock of the
                    #             timed region, against the chip's issue rate: every SIMD starts one such instruction per 2 clocks.
                    #  line_rate: BVH nodes + primitive records the traversal kernels fetch per second of wall clock, against what the
                    #             vector L1s of the chip deliver to divergent lane addresses with the tree resident in L2
                    #             (tools/ubench_chase.hip, profiles/r2_ubench_chase.txt: 165-174 G 64-byte node visits per second).
                    prop = torch.cuda.get_device_properties(0)
                    peak = prop.multi_processor_count*4*MAX_CLOCK_HZ/VALU_CYCLES_PER_WAVE64*1e-9
                    pmc_spp = spp                    # (the pass's own spp: a shorter pass spends a larger share of its wave-instructions in its drain)
                    # two child passes: the instruction total with its f32 / convert / integer categories, then the f64 ones
                    cats_a = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                              "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"]
                    cats_b = ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"]
                    va = measure_counters_all(a, scene, w, h, pmc_spp, self.tmp, cats_a, "a")
                    vb = measure_counters_all(a, scene, w, h, pmc_spp, self.tmp, cats_b, "b") if va else None
                    # enabled lanes per issued VALU instruction (counter_defs.yaml: VALUUtilization; calibrated by tools/ubench_lanes.hip)
                    vc = measure_counters_all(a, scene, w, h, max(4, pmc_spp//8), self.tmp, ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"], "c") if va else None
                    if va:
                        total = sum(c.get("SQ_INSTS_VALU", 0.0) for c in va.values())
                        per_sample = total/float(w*h*pmc_spp)
                        ach = per_sample*value*1e6*1e-9
                        roofline["valu"] = {"bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                                            "frac": round(ach/peak, 4), "instructions_per_sample": round(per_sample, 1),
                                            "share": {k: round(c.get("SQ_INSTS_VALU", 0.0)/total, 3) for k, c in sorted(va.items())},
                                            "source": "this run: rocprofv3 --pmc SQ_INSTS_VALU + categories (two child passes at %d spp) x the timed region's samples/s; "
                                                      "peak = CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md: v_fma_f32 = 2 cyc "
                                                      "on the 32-lane SIMD; max clock, sustained clocks are lower, so frac understates)" % pmc_spp}
                        if vc:
                            tc = sum(c.get("SQ_THREAD_CYCLES_VALU", 0.0) for c in vc.values())
                            ta = sum(c.get("SQ_ACTIVE_INST_VALU", 0.0) for c in vc.values())
                            roofline["valu"]["lane_utilisation"] = {
                                "loop": round(tc/(64.0*ta), 4) if ta else None,
                                "per_kernel": {k: round(c["SQ_THREAD_CYCLES_VALU"]/(64.0*c["SQ_ACTIVE_INST_VALU"]), 4)
                                               for k, c in sorted(vc.items()) if c.get("SQ_ACTIVE_INST_VALU")},
                                "source": "this run: rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU (one child pass at %d spp): enabled lanes per issued "
                                          "VALU instruction / 64 (rocprofiler-sdk's VALUUtilization; the counter pair is calibrated by tools/ubench_lanes.hip, "
                                          "profiles/r6_ubench_lanes.txt)" % max(4, pmc_spp//8)}
                        if walk_stats:
                            roofline["valu"]["walk"] = walk_stats
                        # ... and priced: every category at what a wave64 instruction of it occupies its SIMD for (VALU_PRICE above, measured).
                        # SIMD-cycles per sample x samples/s against SIMDs x clock: how much of the chip's VALU time the loop uses.
                        lo = hi = 0.0
                        mix, share_lo = {}, {}
                        for k, c in va.items():
                            cc2 = dict(c)
                            cc2.update((vb or {}).get(k, {}))
                            known = sum(v2 for n, v2 in cc2.items() if n != "SQ_INSTS_VALU" and n in VALU_PRICE)
                            cc2["other"] = max(cc2.get("SQ_INSTS_VALU", 0.0) - known, 0.0)
                            klo = sum(VALU_PRICE[n][0]*v2 for n, v2 in cc2.items() if n in VALU_PRICE)
                            khi = sum(VALU_PRICE[n][1]*v2 for n, v2 in cc2.items() if n in VALU_PRICE)
                            lo += klo; hi += khi
                            share_lo[k] = klo
                            for n, v2 in cc2.items():
                                if n in VALU_PRICE:
                                    mix[n] = mix.get(n, 0.0) + v2
                        simd_hz = prop.multi_processor_count*4*MAX_CLOCK_HZ
                        cyc_lo, cyc_hi = lo/float(w*h*pmc_spp), hi/float(w*h*pmc_spp)
                        # ONE number instead of the interval: every instantiation's instruction count at the average price of ITS OWN instruction
                        # mix, read off the library's code object (tools/isa_histogram.py: the kernel's main loop, every opcode at its measured cost)
                        static = None
                        try:
                            sys.path.insert(0, os.path.join(ROOT, "tools"))
                            import isa_histogram
                            inst = LAST_VARIANTS.get("a", {})
                            hist = isa_histogram.kernel_prices(tg.capi.LIB_PATH, sorted(set(re.sub(r"<.*", "", v) for v in inst)))
                            cyc, per_variant, missing = 0.0, {}, []
                            for v, c in inst.items():
                                n = c.get("SQ_INSTS_VALU", 0.0)
                                hv = hist.get(v)
                                p = (hv["loop"]["cycles_per_instruction"] or hv["kernel"]["cycles_per_instruction"]) if hv else None
                                if p is None:
                                    missing.append(v)
                                    p = 2.5
                                cyc += n*p
                                if n/total >= 0.005:
                                    per_variant[v] = {"share_of_instructions": round(n/total, 4), "cycles_per_instruction": p}
                            static = {"simd_cycles_per_sample": round(cyc/float(w*h*pmc_spp), 1),
                                      "cycles_per_instruction": round(cyc/total, 3),
                                      "frac_of_simd_time_at_max_clock": round(cyc/float(w*h*pmc_spp)*value*1e6/simd_hz, 4),
                                      "per_kernel": per_variant, "not_found": missing,
                                      "source": "tools/isa_histogram.py on %s: VALU opcodes of each kernel's main loop x tools/ubench_valu.hip's cycles "
                                                "(profiles/r5_ubench_valu.txt, r6_ubench_valu.txt), weighted by this run's SQ_INSTS_VALU per instantiation" % os.path.basename(tg.capi.LIB_PATH)}
                        except Exception as e:     # (objdump missing on a box: the interval below stays)
                            static = {"error": str(e)}
                        roofline["valu"]["priced"] = {
                            "static_mix": static,
                            "simd_cycles_per_sample": [round(cyc_lo, 1), round(cyc_hi, 1)],
                            "frac_of_simd_time_at_max_clock": [round(cyc_lo*value*1e6/simd_hz, 4), round(cyc_hi*value*1e6/simd_hz, 4)],
                            "mix": {n.replace("SQ_INSTS_VALU_", "").lower(): round(v2/total, 4) for n, v2 in sorted(mix.items())},
                            "share": {k: round(v2/lo, 3) for k, v2 in sorted(share_lo.items())},
                            "prices": "cycles per wave64 instruction per SIMD, low / high, tools/ubench_valu.hip -> profiles/r5_ubench_valu.txt (v_fma_f32 = 2): "
                                      + ", ".join("%s %.1f-%.1f" % (n.replace("SQ_INSTS_VALU_", "").lower(), p0, p1) for n, (p0, p1) in sorted(VALU_PRICE.items())),
                            "note": "low prices INT32 and the uncategorised rest (moves, f32 compares / selects / min / max, lane ops) as 2-cycle instructions, high "
                                    "as 3.2-cycle ones; the truth is in between.  Sustained clocks under this load are below 2.4 GHz, so both fractions understate."}
                    visits = (cc["nodes_visited"] + cc["prims_tested"])*steps/elapsed*1e-9
                    roofline["line_rate"] = {"bound": "l1-line-rate", "achieved": round(visits, 1), "peak": 170.0, "unit": "G node+record visits/s",
                                             "frac": round(visits/170.0, 4),
                                             "note": "visits of the whole loop over its wall clock (the traversal kernels hold the chip for about half "
                                                     "of it, next to the shading launches of the other parts); peak = L2-resident divergent walk, "
                                                     "tools/ubench_chase.hip; 67 G/s once the tree spills to the Infinity Cache"}
                if (a.traffic or getattr(a, "valu_pass", False)) and self.world == 1:
                    if fused:
                        # what this kernel IS bound by: VALU issue.  One more counter pass (SQ_INSTS_VALU per launch) against the chip's
                        # issue rate: every SIMD starts one wave-wide VALU instruction per 2 clocks (32 lanes x 2 = 64).
                        prop = torch.cuda.get_device_properties(0)
                        peak = prop.multi_processor_count*4*MAX_CLOCK_HZ/VALU_CYCLES_PER_WAVE64*1e-9
                        v = measure_counter(a, scene, w, h, spp, dom, self.tmp, "SQ_INSTS_VALU")
                        if v is not None:
                            ach = v/(kd["avg_us"]*1e-6)*1e-9
                            roofline["valu"] = {"bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                                                "frac": round(ach/peak, 4), "instructions_per_launch": round(v),
                                                "source": "this run: rocprofv3 --pmc SQ_INSTS_VALU (one child pass); peak = CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 "
                                                          "instruction (MI355X_MICROARCH.md: v_fma_f32 = 2 cyc on the 32-lane SIMD; max clock, sustained clocks are lower)"}
                            vc = measure_counters_all(a, scene, w, h, spp, self.tmp, ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"], "c")
                            if vc and vc.get(dom, {}).get("SQ_ACTIVE_INST_VALU"):
                                roofline["valu"]["lane_utilisation"] = round(vc[dom]["SQ_THREAD_CYCLES_VALU"]/(64.0*vc[dom]["SQ_ACTIVE_INST_VALU"]), 4)
                if self.world == 1 and not fused and parts > 1 and a.exclusive and not a.no_kernel_timing and not a.emulate_shards:
                    # The same kernels with the chip to themselves: the pool as ONE part on one stream ("streams" = 1), so that no launch
                    # overlaps another and, per class, average launch time x launches <= the wall clock of the region (checked below).
                    # Slower as a whole -- nothing fills the drain of a launch -- but each launch's duration is its own.
                    check(lib.tghip_set_option(ctx, b"streams", 1), "tghip_set_option")
                    x_steps = 2
                    step()
                    check(lib.tghip_reset_counters(ctx), "tghip_reset_counters")
                    check(lib.tghip_set_option(ctx, b"time_kernels", 1), "tghip_set_option")
                    self.fence()
                    tx = time.perf_counter()
                    for _ in range(x_steps):
                        step()
                    self.fence()
                    x_elapsed = time.perf_counter() - tx
                    xt = tg.TgHipCounters()
                    lib.tghip_get_counters(ctx, C.byref(xt))
                    xt = counters_dict(xt)
                    check(lib.tghip_set_option(ctx, b"time_kernels", 0), "tghip_set_option")
                    restore = [int(kv.split("=")[1]) for kv in a.opt if kv.split("=")[0] == "streams"]
                    check(lib.tghip_set_option(ctx, b"streams", restore[-1] if restore else 0), "tghip_set_option")
                    xms = {"k_trace_closest": xt["ms_trace_closest"], "k_shade": xt["ms_shade"], "k_trace_shadow": xt["ms_trace_shadow"]}
                    xl = {"k_trace_closest": xt["launches_trace_closest"], "k_shade": xt["launches_shade"], "k_trace_shadow": xt["launches_trace_shadow"]}
                    ex = {"value": round(float(w)*h*spp*x_steps/x_elapsed*1e-6, 2), "unit": "Msamples/s", "steps": x_steps,
                          "ms_per_step": round(x_elapsed/x_steps*1e3, 3), "iterations": int(xt["iterations"]), "per_kernel": {},
                          "note": "one part on one stream (option streams=1): every launch has the chip to itself; same byte model, same counts"}
                    for k in per_step_bytes:
                        if xl[k] and xms[k] > 0:
                            bpl = per_step_bytes[k]*x_steps/xl[k]
                            avg = xms[k]*1e-3/xl[k]
                            ex["per_kernel"][k] = {"achieved": round(bpl/avg*1e-9, 1), "frac": round(bpl/avg*1e-9/HBM_PEAK_GBS, 4), "bytes_per_launch": round(bpl),
                                                   "avg_launch_us": round(avg*1e6, 2), "launches": int(xl[k]),
                                                   "share_of_wall": round(xms[k]*1e-3/x_elapsed, 4)}
                    ex["launch_time_over_wall"] = round(sum(xms.values())*1e-3/x_elapsed, 4)     # <= 1: the launches do not overlap
                    if dom_trav in ex["per_kernel"]:
                        ex.update({"kernel": dom_trav, "achieved": ex["per_kernel"][dom_trav]["achieved"], "frac": ex["per_kernel"][dom_trav]["frac"],
                                   "avg_launch_us": ex["per_kernel"][dom_trav]["avg_launch_us"]})
                    roofline["exclusive"] = ex
            if sustained_clock and isinstance(roofline.get("valu"), dict) and roofline["valu"].get("frac"):
                # the VALU ceiling at the clock the chip actually ran at during the timed region (the line's `peak` is priced at the maximum clock)
                roofline["valu"]["frac_at_sustained_clock"] = round(roofline["valu"]["frac"]/sustained_clock["frac_of_max_clock"], 4)
                st = (roofline["valu"].get("priced") or {}).get("static_mix") or {}
                if st.get("frac_of_simd_time_at_max_clock"):
                    st["frac_of_simd_time_at_sustained_clock"] = round(st["frac_of_simd_time_at_max_clock"]/sustained_clock["frac_of_max_clock"], 4)
            rays = max(cc["closest_rays"] + cc["shadow_rays"], 1)
            out = {
                "value": round(value, 2), "ms_per_step": round(elapsed/steps*1e3, 3),
                "config": {"workload": workload, "width": w, "height": h, "spp": spp, "sampler": "uniform (counter-based PCG)",
                           "adaptive_sampling": False, "max_bounces": int(flat.desc.contents.settings.max_bounces),
                           "parallelism": "tile-shard x%d%s" % (self.world, (" + framebuffer reduce: " + reduce_note) if self.world > 1 else "")},
                "cpu_baseline": cpu_baseline(a, scene, path, flat, w, h, spp, self.tmp) if cpu else None,
                "roofline": ordered_roofline(roofline),
                "sustained_clock": sustained_clock,
                "kernels": kernels,
                "count_pass_spp": count_spp,
                "walk": walk_stats or None,        # (also under roofline.valu.walk when the counter passes ran)
                "rays_per_sample": round(rays/max(cc["samples"], 1), 3),
                "nodes_per_ray": round(cc["nodes_visited"]/rays, 2), "prims_per_ray": round(cc["prims_tested"]/rays, 2),
                "bvh": {"nodes": int(flat.info.num_nodes), "records": int(flat.info.num_recs), "depth": int(flat.info.bvh_depth),
                        "flat_list": is_flat, "wide_nodes": int(flat.desc.contents.num_wide_nodes) if wide else 0},
                "kernel_ms_total": round(timed["ms_total"], 2), "wavefront_iterations": int(timed["iterations"]),
                "setup_s": {"flatten_and_bvh": round(t_flatten, 3), "upload": round(t_upload, 3)},
                "result_ok": ok, "image_mean": [round(float(v), 6) for v in img.mean(axis=(0, 1))],
            }
            if emulated:
                out["emulated_shards"] = emulated
            if per_rank:
                out["per_rank"] = per_rank
            else:
                out["render_ms_per_step"] = round(split[0]/steps*1e3, 3)
        lib.tghip_bind_framebuffer(ctx, None, None)
        if not shared:
            lib.tghip_destroy(ctx)
        flat.close()
        return out


def run_in_process(b, a, scene, w, h, spp):
    """--in-process: N GPUs driven by ONE process through the host integrator (\"devices\": N -- N contexts on N host threads, each rendering its tile
    shard of every pass, tghip_reduce_framebuffers = ncclReduce into device 0, download of the merged image): the product's own multi-GPU path end
    to end, as a Tungsten front end would call it.  A step = tgh_renderer_render of a freshly opened renderer (scene load and upload untimed)."""
    import numpy as np
    tg = b.tg
    if tg.device_count() < a.gpus:
        raise SystemExit("bench.py --in-process: %d devices visible, --gpus %d" % (tg.device_count(), a.gpus))
    path, workload = b.make_scene(scene, w, h, spp)
    secs, ok, img_mean, per_device = [], True, None, None
    for i in range(a.warmup + a.steps):
        r = tg.Renderer(path, devices=a.gpus)
        t = r.render()
        mean, _, count = r.image()
        if i >= a.warmup:
            secs.append(t)
            ok = ok and bool((count == spp).all() and np.isfinite(mean).all())
            img_mean = [round(float(v), 6) for v in mean.mean(axis=(0, 1))]
            per_device = [int(r.counters(d).samples) for d in range(a.gpus)]
        r.close()
    total = sum(secs)
    return {"value": round(float(w)*h*spp*len(secs)/total*1e-6, 2), "ms_per_step": round(total/len(secs)*1e3, 3),
            "config": {"workload": workload, "width": w, "height": h, "spp": spp, "sampler": "uniform (counter-based PCG)", "adaptive_sampling": False,
                       "parallelism": "tile-shard x%d in one process: host integrator \"devices\": %d + tghip_reduce_framebuffers (ncclReduce)" % (a.gpus, a.gpus)},
            "roofline": None, "cpu_baseline": None, "result_ok": ok, "image_mean": img_mean,
            "per_rank": {"samples_rendered_by_device": per_device, "render_s_per_step": [round(t, 4) for t in secs],
                         "note": "one process: the step's wall clock covers every device's shard, the reduce and the download of the merged image"}}


def kernel_class(name):
    """Kernel class of a demangled kernel name as rocprofv3 prints it ("void k_finish_trace_closest_wide<1u, 0, true>(PassParams, ...)"):
    the key of kernel_bytes() the launch is priced under.  k_finish_trace_closest_wide is the closest-hit walk with the previous
    iteration's finish work folded in front -- class k_trace_closest; the _dyn / _wide / _fast suffixes name walks."""
    k = re.sub(r"<.*", "", re.sub(r"\(.*", "", name).replace("void ", "").strip())
    if k.startswith("k_finish_trace_closest") or k.startswith("k_trace_closest"):
        return "k_trace_closest"
    if k.startswith("k_trace_shadow"):
        return "k_trace_shadow"
    if k.startswith("k_shade"):
        return "k_shade"
    return k


def measure_traffic(a, scene, w, h, spp, tmp):
    """HBM bytes per launch of every kernel class, measured in THIS run: two child runs of this script under `rocprofv3 --pmc`
    (FETCH_SIZE and WRITE_SIZE need separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots") on the same scene at a
    fraction of the spp (counter collection serialises dispatches), then (2*FETCH_SIZE + WRITE_SIZE)*1024: KiB units, read
    side doubled on gfx950 as MI355X_MICROARCH.md "HBM" prescribes.  Returns ({class: bytes per launch}, source string);
    ({}, reason) on failure."""
    import csv
    exe = shutil.which("rocprofv3")
    if not exe:
        return {}, "rocprofv3 not found"
    pmc_spp = max(4, spp//8) if scene != "cornell" else spp
    per_class = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = os.path.join(tmp, "pmc_" + counter)
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--scene", scene, "--material", a.material, "--res", "%dx%d" % (w, h), "--spp", str(pmc_spp), "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline", "--no-extra", "--no-kernel-timing", "--no-traffic", "--no-clock"] + [x for kv in a.opt for x in ("--opt", kv)]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd="/tmp", env=env, timeout=300)
        except Exception as e:
            return {}, "rocprofv3 --pmc %s failed: %s" % (counter, e)
        files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
        if p.returncode != 0 or not files:
            return {}, "rocprofv3 --pmc %s: rc %d, no counter file" % (counter, p.returncode)
        sums, ids = {}, {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = kernel_class(row["Kernel_Name"])
                sums[k] = sums.get(k, 0.0) + float(row["Counter_Value"])
                ids.setdefault(k, set()).add(row.get("Dispatch_Id"))
        shutil.rmtree(out, ignore_errors=True)
        if not sums:
            return {}, "rocprofv3 --pmc %s: the counter file holds no %s rows" % (counter, counter)
        per_class[counter] = {k: (sums[k]/len(ids[k]), len(ids[k])) for k in sums}
    traffic = {}
    for k, (f, n) in per_class["FETCH_SIZE"].items():
        if k in per_class["WRITE_SIZE"]:
            traffic[k] = {"bytes_per_launch": round((2.0*f + per_class["WRITE_SIZE"][k][0])*1024.0), "launches": n}
    return traffic, ("this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two child passes at %d spp), (2*FETCH + WRITE) KiB per "
                     "MI355X_MICROARCH.md" % pmc_spp)


def measure_counter_all(a, scene, w, h, spp, tmp, counter):
    """One PMC counter summed per kernel class over one child run under rocprofv3 (all rows of every dispatch added up):
    {kernel class: (total, dispatches)}; None on failure."""
    import csv
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = os.path.join(tmp, "pmc_all_" + counter)
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
           "--scene", scene, "--material", a.material, "--res", "%dx%d" % (w, h), "--spp", str(spp), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-extra", "--no-kernel-timing", "--no-traffic", "--no-clock"] + [x for kv in a.opt for x in ("--opt", kv)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), timeout=300)
    except Exception:
        return None
    files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
    if p.returncode != 0 or not files:
        return None
    sums, ids = {}, {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            k = kernel_class(row["Kernel_Name"])
            if row.get("Counter_Name") != counter or not k.startswith("k_"):
                continue
            sums[k] = sums.get(k, 0.0) + float(row["Counter_Value"])
            ids.setdefault(k, set()).add(row.get("Dispatch_Id"))
    shutil.rmtree(out, ignore_errors=True)
    return {k: (v, len(ids[k])) for k, v in sums.items()} or None


LAST_VARIANTS = {}     # measure_counters_all: the same sums per kernel INSTANTIATION (template arguments kept), by pass tag


def measure_counters_all(a, scene, w, h, spp, tmp, counters, tag):
    """Several PMC counters (one pass: at most eight SQ counters fit, MI355X_MICROARCH.md "rocprofv3 PMC slots") summed per kernel class over
    one child run under rocprofv3: {kernel class: {counter: total}}; None on failure."""
    import csv
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = os.path.join(tmp, "pmc_multi_" + tag)
    cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
           "--scene", scene, "--material", a.material, "--res", "%dx%d" % (w, h), "--spp", str(spp), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-extra", "--no-kernel-timing", "--no-traffic", "--no-clock"] + [x for kv in a.opt for x in ("--opt", kv)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), timeout=300)
    except Exception:
        return None
    files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
    if p.returncode != 0 or not files:
        return None
    sums, variants = {}, {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            k = kernel_class(row["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            e = sums.setdefault(k, {})
            e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            v = variants.setdefault(re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip(), {})      # template arguments kept
            v[row["Counter_Name"]] = v.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    shutil.rmtree(out, ignore_errors=True)
    LAST_VARIANTS[tag] = variants
    return sums or None


def measure_counter(a, scene, w, h, spp, kernel, tmp, counter):
    """Sum of one PMC counter per launch of `kernel` (all rows of a dispatch added up), from one child run under rocprofv3; None on failure."""
    import csv
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = os.path.join(tmp, "pmc_" + counter)
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
           "--scene", scene, "--material", a.material, "--res", "%dx%d" % (w, h), "--spp", str(spp), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-extra", "--no-kernel-timing", "--no-traffic", "--no-clock"]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), timeout=300)
    except Exception:
        return None
    files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
    if p.returncode != 0 or not files:
        return None
    total, ids = 0.0, set()
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == counter and kernel_class(row["Kernel_Name"]) == kernel:
                total += float(row["Counter_Value"])
                ids.add(row.get("Dispatch_Id"))
    shutil.rmtree(out, ignore_errors=True)
    return total/len(ids) if ids else None


def cpu_baseline(a, scene, path, flat, w, h, spp, tmp):
    """The reference's own CPU/Embree path on all host cores, same scene at the same resolution with fewer spp (bounded
    sample): oracle/_ref/avx2/tungsten (the haswell / AVX2-Embree build SURVEY.md 8d asks for) or, failing that, the default
    SSE4.2 build oracle/_ref/tungsten; median of 3 runs, each timed by the wall clock between the binary's "Starting
    render..." and "Finished render" log lines (its own "Render time" has 1-second granularity).  Falls back to the oracle
    port when no reference binary is present."""
    import tungsten_amd as tg
    cores = os.cpu_count() or 1
    refs = [(os.path.join(ROOT, "oracle", "_ref", "avx2", "tungsten"), "haswell/AVX2-Embree build"),
            (os.path.join(ROOT, "oracle", "_ref", "tungsten"), "SSE4.2 build (reference CMake default)")]
    # rough CPU rates (Msamples/s per core) to size the sample: cornell ~0.7, materialtest ~0.3; the reference stops
    # scaling long before 256 threads (its tile pool), so cap the estimate at 32 cores' worth
    per_core = 0.7 if scene == "cornell" else 0.3 if scene == "materialtest" else 0.15
    budget = a.cpu_seconds/6.0*per_core*min(cores, 32)*1e6
    s_spp = int(max(1, min(spp, budget//(w*h))))
    # the unmodified reference binary never loads the mesh files of an Instance's masters (Instance.cpp:265-282), i.e. it
    # would render the instanced scene without its instances: that scene is timed with the oracle port instead
    results = []
    for ref, build in refs:
        if scene == "instances10k" or not (os.path.exists(ref) and os.access(ref, os.X_OK)):
            continue
        times = []
        try:
            for run in range(3):
                p = subprocess.Popen([ref, "-t", str(cores), "-s", str(tg.DEFAULT_SEED), "--spp", str(s_spp),
                                      "-e", os.path.join(tmp, "cpu_%s.pfm" % scene), "-o", os.path.join(tmp, "cpu_%s.png" % scene), path],
                                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd=tmp, bufsize=1)
                t_start = t_end = None
                for line in p.stdout:
                    if "Starting render" in line:
                        t_start = time.perf_counter()
                    elif "Finished render" in line:
                        t_end = time.perf_counter()
                p.wait(timeout=600)
                if p.returncode != 0 or t_start is None or t_end is None:
                    raise RuntimeError("reference binary failed")
                times.append(t_end - t_start)
        except Exception:
            continue
        times.sort()
        results.append((w*h*s_spp/times[1]*1e-6, build, times))
    if results:
        # both builds are timed (the AVX2 one is not always the faster one); the baseline is the better median
        best = max(results, key=lambda r: r[0])
        return {"value": round(best[0], 3), "unit": "Msamples/s", "cores": cores, "kind": "reference",
                "sample": "%dx%d @ %d spp of the same scene, reference binary `tungsten -t %d`, render-loop wall clock, median of 3 runs per build: "
                          % (w, h, s_spp, cores) + "; ".join("%s %s s => %.2f Msamples/s" % (b, " / ".join("%.2f" % t for t in ts), v)
                                                              for v, b, ts in results)}
    sys.path.insert(0, os.path.join(ROOT, "tests"))      # (the checker's ctypes wrapper lives with the tests; only this fallback leg uses it)
    import oracle_lib
    s_spp = max(1, s_spp//2)
    t0 = time.time()
    oracle_lib.render(flat.desc, w, h, 0, s_spp, tg.DEFAULT_SEED, threads=cores)
    secs = time.time() - t0
    return {"value": round(w*h*s_spp/secs*1e-6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%dx%d @ %d spp of the same scene, oracle/oracle.c with OpenMP on %d threads, %.2f s" % (w, h, s_spp, cores, secs)}


def main():
    a = parse_args()
    from tungsten_amd import workloads as scenes
    b = Bench(a)
    try:
        w, h = [int(v) for v in a.res.split("x")]
        scene = a.scene
        if scene in ("materialtest", "mesh1m") and not scenes.have_materialtest():
            # the headline workload is materialtest: a run without its assets is not a measurement of the metric (no silent fallback)
            raise SystemExit("bench.py: materialtest assets (assets/materialtest) missing -- run __graft_entry__.build() where the "
                             "reference is mounted, or pass --scene cornell explicitly")
        spp = a.spp or (256 if scene in ("cornell", "materialtest") else 32)
        if scene in ("mesh1m", "instances10k") and a.res == "1280x720":
            w, h = 1920, 1080
        cpu = not a.no_cpu_baseline and b.world == 1
        in_process = a.in_process and a.gpus > 1
        res = run_in_process(b, a, scene, w, h, spp) if in_process else b.run(scene, w, h, spp, a.steps, a.warmup, cpu)
        extra = None
        if scene == "materialtest" and b.world == 1 and not a.no_extra and not in_process:
            # BASELINE configs[1] on the same line: a flat-list scene, no traversal kernel (2 steps)
            saved, a.traffic = a.traffic, False
            a.valu_pass = saved                  # (no HBM traffic passes for the extra line, but its VALU-issue roofline)
            m = b.run("cornell", 1280, 720, 256, 2, 1, False)
            a.traffic, a.valu_pass = saved, False
            if b.rank == 0:
                keys = ("value", "ms_per_step", "config", "roofline", "kernels", "rays_per_sample", "prims_per_ray", "bvh", "result_ok")
                extra = {"cornell_1280x720_256spp": dict({k: m[k] for k in keys}, unit="Msamples/s", steps=2, warmup=1)}
                # and the metric's scene AS THE REFERENCE SHIPS IT: Sobol' sampler + adaptive sampling in 16-spp passes through the whole
                # integrator loop (host scheduler between the passes included), 64 spp, best of three renderers (tools/bench_as_shipped.py)
                try:
                    extra["materialtest_as_shipped_1280x720_64spp"] = as_shipped(b.tmp, tg_mod=b.tg)
                except Exception as e:       # (never takes the headline line down)
                    extra["materialtest_as_shipped_1280x720_64spp"] = {"error": str(e)}
        if b.rank == 0:
            out = {"metric": "Msamples/s (W*H*spp/s), path_tracer render loop", "value": res["value"], "unit": "Msamples/s",
                   "n_gpus": a.gpus if in_process else b.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"],
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                   "data": ("the reference's shipped scene data/materialtest/materialtest.json (meshes, HDRI and materials as shipped; copied by build() into "
                            "assets/), resolution / spp / sampler set by the bench" if scene == "materialtest" else
                            "scene description generated in-tree (tungsten_amd/workloads.py: %s)" % scene) + "; fixed seed 0xBA5EBA11"}
            out.update({k: v for k, v in res.items() if k not in ("value", "ms_per_step")})
            if extra:
                out["extra"] = extra
            print(json.dumps(out))
            if not res["result_ok"]:
                raise SystemExit("bench.py: rendered image failed the sanity check")
    finally:
        b.close()


if __name__ == "__main__":
    main()
