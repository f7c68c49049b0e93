"""Life cycle of the C-ABI's objects on the device: contexts, scenes and pools created and destroyed many times must give their memory back.  A front end that re-opens
a renderer per frame (the reference's editor and its batch mode both do: one TraceableScene per render, renderer/TraceableScene.hpp:64-134) would otherwise run a long
session out of HBM."""
import ctypes as C

import numpy as np
import pytest

import scenes
import tungsten_amd as tg

pytestmark = pytest.mark.gpu


def _free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_renderers_give_their_memory_back(tmp_path):
    """60 renders, each through a renderer of its own (scene upload, path pool, framebuffer, adaptive-pass buffers, auxiliary buffers in every third one):
    the device's free memory after the 60th is what it was after the 10th, and every render is the first one's image."""
    plain = scenes.cornell(tmp_path, name="a.json", resolution=(160, 90), spp=8)
    adaptive = scenes.cornell(tmp_path, name="b.json", resolution=(160, 90), spp=32, spp_step=16, renderer={"adaptive_sampling": True, "stratified_sampler": True})
    first, marks = {}, {}
    for i in range(60):
        path = adaptive if i % 3 == 2 else plain
        r = tg.Renderer(path, seed=tg.DEFAULT_SEED)
        r.render()
        mean = r.image()[0]
        r.close()
        if path in first:
            assert np.array_equal(mean, first[path]), "render %d differs from the first of its scene" % i
        else:
            first[path] = mean
        if i in (9, 59):
            marks[i] = _free_bytes()
    assert marks[9] - marks[59] < 32 << 20, "device memory shrank by %.1f MB over 50 renderers" % ((marks[9] - marks[59])/2.0**20)
