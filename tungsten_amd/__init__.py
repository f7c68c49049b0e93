"""tungsten_amd -- MI355X-native `path_tracer_hip` integrator behind Tungsten's plugin surface.

Python here is only a ctypes convenience layer over the C-ABI (include/tungsten_hip.h,
include/tungsten_host.h) for tests and bench.py; all rendering happens in the native library
(C++11 host + HIP kernels for gfx950).  Importing the package fails if that library is missing:
there is no Python or CPU fallback.

A process that also uses PyTorch-ROCm must `import torch` BEFORE this package: torch ships its own HIP / HSA runtime, and two copies in one
process do not share the device (bench.py and the tests' multi-process workers import torch first; nothing here imports it).
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import (TgHipCounters, TgHipHit, TgHipPassDesc, TgHipRay, TgHipSceneDesc, TgHostSceneInfo)

lib = capi.load_library()

AUX_DTYPE = np.dtype([("a", np.float32, 11), ("b", np.float32, 11), ("variance", np.float32, 11), ("count", np.uint32, 5)])
RECORD_DTYPE = np.dtype([("sample_count", np.uint32), ("next_sample_count", np.uint32), ("sample_index", np.uint32),
                         ("adaptive_weight", np.float32), ("mean", np.float32), ("running_variance", np.float32)])
DEFAULT_SEED = 0xBA5EBA11  # src/tungsten/Shared.hpp:246


class TungstenError(RuntimeError):
    pass


def device_count():
    return int(lib.tghip_device_count())


class FlattenedScene(object):
    """Scene::load + loadResources + TraceableScene flattening (no device needed)."""

    def __init__(self, json_path):
        err = C.create_string_buffer(1024)
        self._h = lib.tgh_scene_load(os.fsencode(json_path), err, len(err))
        if not self._h:
            raise TungstenError(err.value.decode(errors="replace"))
        self.desc = lib.tgh_scene_desc(self._h)
        info = TgHostSceneInfo()
        lib.tgh_scene_info(self._h, C.byref(info))
        self.info = info

    @property
    def width(self):
        return int(self.info.width)

    @property
    def height(self):
        return int(self.info.height)

    def items(self):
        """The items of the reference's top-level Embree geometry -- the scene's finite primitives in scene order -- as (boxes [n, 6] float32:
        each one's bounds(), object indices [n] int32)."""
        boxes, objects = C.POINTER(C.c_float)(), C.POINTER(C.c_int32)()
        n = lib.tgh_scene_items(self._h, C.byref(boxes), C.byref(objects))
        if not n:
            return np.zeros((0, 6), np.float32), np.zeros(0, np.int32)
        return (np.ctypeslib.as_array(boxes, shape=(n, 6)).copy(), np.ctypeslib.as_array(objects, shape=(n,)).copy())

    def close(self):
        if self._h:
            lib.tgh_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer(object):
    """makeTraceable(seed) with the path_tracer_hip integrator + the CLI render loop."""

    def __init__(self, json_path, seed=DEFAULT_SEED, spp=0, devices=0):
        err = C.create_string_buffer(1024)
        self._err = err
        self._h = lib.tgh_renderer_open(os.fsencode(json_path), seed & 0xFFFFFFFF, int(spp), int(devices), err, len(err))
        if not self._h:
            raise TungstenError(err.value.decode(errors="replace"))
        info = TgHostSceneInfo()
        lib.tgh_renderer_info(self._h, C.byref(info))
        self.info = info
        self.width, self.height = int(info.width), int(info.height)

    def _check(self, rc):
        if rc != 0:
            raise TungstenError(self._err.value.decode(errors="replace"))

    def context(self, device=0):
        return lib.tgh_renderer_context(self._h, device)

    def set_option(self, key, value, device=0):
        rc = lib.tghip_set_option(self.context(device), key.encode(), int(value))
        if rc != 0:
            raise TungstenError(lib.tghip_last_error(self.context(device)).decode())

    def step(self):
        done = C.c_int(0)
        self._check(lib.tgh_renderer_step(self._h, C.byref(done), self._err, len(self._err)))
        return bool(done.value)

    def render(self):
        """while (!done) { startRender; waitForCompletion }; returns wall seconds of the loop."""
        secs = C.c_double(0.0)
        self._check(lib.tgh_renderer_render(self._h, C.byref(secs), self._err, len(self._err)))
        return secs.value

    def image(self):
        """(mean, sum, count): mean radiance [H,W,3], raw radiance sums [H,W,3], sample counts [H,W]."""
        n = self.width*self.height
        mean = np.empty((self.height, self.width, 3), np.float32)
        ssum = np.empty((self.height, self.width, 3), np.float32)
        count = np.empty((self.height, self.width), np.uint32)
        self._check(lib.tgh_renderer_image(self._h, mean.ctypes.data, ssum.ctypes.data, count.ctypes.data, n,
                                           self._err, len(self._err)))
        return mean, ssum, count

    def counters(self, device=0):
        c = TgHipCounters()
        lib.tghip_get_counters(self.context(device), C.byref(c))
        return c

    def reset_counters(self, device=0):
        lib.tghip_reset_counters(self.context(device))

    def trace_rays(self, rays, repeats=1, device=0):
        """Batched TraceableScene::intersect: rays [N,8] float32 (o, tmin, d, tmax) -> hits (t,u,v,rec)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.empty(rays.shape[0], dtype=[("t", np.float32), ("u", np.float32), ("v", np.float32), ("rec", np.int32)])
        ms = C.c_double(0.0)
        rc = lib.tghip_trace_rays(self.context(device), rays.ctypes.data, hits.ctypes.data, rays.shape[0], int(repeats), C.byref(ms))
        if rc != 0:
            raise TungstenError(lib.tghip_last_error(self.context(device)).decode())
        return hits, ms.value

    def debug_libm(self, fn, x, device=0):
        """The device's restatement of a host libm function (capi.TGHIP_LIBM_*) evaluated on the device: float32 in, float32 out."""
        dbl = int(fn) >= capi.TGHIP_LIBM_EXPD                                 # the double-precision functions: float64 in, float64 out
        x = np.ascontiguousarray(x, np.float64 if dbl else np.float32).reshape(-1)
        two = int(fn) in (capi.TGHIP_LIBM_ATAN2F, capi.TGHIP_LIBM_POWF)      # operands interleaved: x[2i], x[2i+1]
        y = np.empty(x.size//2 if two else x.size, x.dtype)
        rc = lib.tghip_debug_libm(self.context(device), int(fn), x.ctypes.data, y.ctypes.data, y.size)
        if rc != 0:
            raise TungstenError(lib.tghip_last_error(self.context(device)).decode())
        return y

    def trace_samples(self, spp_begin, spp_end, seed=DEFAULT_SEED, tile_seeds=None, device=0):
        """One TGHIP_PASS_SAMPLES pass on the device (into a cleared framebuffer): the radiance of every individual sample,
        [H, W, spp_end - spp_begin, 3] -- PathTracer::traceSample's return value per (pixel, sample index).  With tile_seeds
        (one per 16x16 tile) the pass draws from the Sobol' sampler (TGHIP_PASS_SOBOL)."""
        ctx = self.context(device)
        flags = capi.TGHIP_PASS_SAMPLES | (capi.TGHIP_PASS_SOBOL if tile_seeds is not None else 0)
        p = TgHipPassDesc(int(spp_begin), int(spp_end), seed & 0xFFFFFFFF, 0, 1, flags)
        if tile_seeds is not None:
            seeds = np.ascontiguousarray(tile_seeds, np.uint32)
            p.tile_seeds = seeds.ctypes.data_as(C.POINTER(C.c_uint32))
        out = np.empty((self.height, self.width, int(spp_end) - int(spp_begin), 3), np.float32)
        for rc in (lib.tghip_clear_framebuffer(ctx), lib.tghip_render_pass(ctx, C.byref(p)), lib.tghip_wait(ctx),
                   lib.tghip_download_samples(ctx, out.ctypes.data, out.size)):
            if rc != 0:
                raise TungstenError(lib.tghip_last_error(ctx).decode())
        return out

    def save_resume_data(self):
        """Integrator::saveRenderResumeData: writes renderer.resume_render_file."""
        self._check(lib.tgh_renderer_save_resume_data(self._h, self._err, len(self._err)))

    def resume(self):
        """Integrator::resumeRender: True when a saved state of this very scene was found and restored."""
        ok = C.c_int(0)
        self._check(lib.tgh_renderer_resume(self._h, C.byref(ok), self._err, len(self._err)))
        return bool(ok.value)

    @property
    def current_spp(self):
        info = TgHostSceneInfo()
        lib.tgh_renderer_info(self._h, C.byref(info))
        return int(info.current_spp)

    def records(self):
        """The integrator's SampleRecords (one per 4x4 pixels) after the last pass, as a structured array [vh, vw]."""
        vw, vh = (self.width + 3)//4, (self.height + 3)//4
        rec = np.zeros(vw*vh, RECORD_DTYPE)
        self._check(lib.tgh_renderer_records(self._h, rec.ctypes.data, rec.size, self._err, len(self._err)))
        return rec.reshape(vh, vw)

    def output_buffers(self):
        """The auxiliary output buffers (renderer.output_buffers) as a structured array [H, W] of AUX_DTYPE: per output the
        A / B halves, the Welford variance sum and the sample count (include/tungsten_hip.h: TgHipAuxPixel)."""
        aux = np.zeros(self.width*self.height, AUX_DTYPE)
        self._check(lib.tgh_renderer_output_buffers(self._h, aux.ctypes.data, aux.size, self._err, len(self._err)))
        return aux.reshape(self.height, self.width)

    def save_outputs(self):
        self._check(lib.tgh_renderer_save_outputs(self._h, self._err, len(self._err)))

    def close(self):
        if self._h:
            lib.tgh_renderer_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PassScheduler(object):
    """PathTraceIntegrator's pass scheduling without a device: tile seeds + generateWork over SampleRecords."""

    def __init__(self, width, height, seed=DEFAULT_SEED):
        self._h = lib.tgh_scheduler_create(int(width), int(height), seed & 0xFFFFFFFF)
        self.num_tiles = int(lib.tgh_scheduler_num_tiles(self._h))
        self.num_records = int(lib.tgh_scheduler_num_records(self._h))

    @property
    def tile_seeds(self):
        return np.ctypeslib.as_array(lib.tgh_scheduler_tile_seeds(self._h), shape=(self.num_tiles,)).copy()

    @property
    def records(self):
        """Mutable view of the records (write sample_count / mean / running_variance, read the rest)."""
        buf = (C.c_char*(self.num_records*RECORD_DTYPE.itemsize)).from_address(
            C.addressof(lib.tgh_scheduler_records(self._h).contents))
        return np.frombuffer(buf, RECORD_DTYPE)

    def generate_work(self, current_spp, next_spp, adaptive):
        return bool(lib.tgh_scheduler_generate_work(self._h, int(current_spp), int(next_spp), int(bool(adaptive))))

    def close(self):
        if self._h:
            lib.tgh_scheduler_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sobol_matrices():
    """The 1024 x 52 Sobol' generator matrices the host hands to the device."""
    n = C.c_size_t(0)
    err = C.create_string_buffer(1024)
    p = lib.tgh_sobol_matrices(C.byref(n), err, len(err))
    if not p:
        raise TungstenError(err.value.decode(errors="replace"))
    return np.ctypeslib.as_array(p, shape=(n.value,)).reshape(1024, 52)


def load_pfm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        w, h = [int(v) for v in f.readline().split()]
        scale = float(f.readline())
        ch = 3 if magic == b"PF" else 1
        data = np.frombuffer(f.read(w*h*ch*4), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, ch)
    return np.ascontiguousarray(data[::-1]).astype(np.float32)
