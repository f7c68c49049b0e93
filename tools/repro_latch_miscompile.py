#!/usr/bin/env python3
"""Reproduces the loop-latch miscompile of the two-level wide shadow walk (pt_wavefront.h: PT_TURN_JOIN) on an MI355X.

Renders instances10k (10 000 instances of a 19 800-triangle mesh) with one pass through four shadow kernels and compares the images:
the two-level BVH2 walk (reference), k_trace_shadow_wide<false, ., INST> as shipped (with the join), the same kernel counting its visits,
and the kernel WITHOUT the join ("inst_shadow_join" = 0) -- twice, because that one differs from run to run.

    python tools/repro_latch_miscompile.py [--res 1920x1080] [--spp 2]
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import tungsten_amd as tg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="1920x1080")
    ap.add_argument("--spp", type=int, default=2)
    a = ap.parse_args()
    w, h = (int(v) for v in a.res.split("x"))
    path = scenes.instances10k(tempfile.mkdtemp(prefix="latch_"), resolution=(w, h), spp=a.spp)
    r = tg.Renderer(path, seed=1234)
    ctx = r.context()

    def render(**opts):
        for k, v in opts.items():
            r.set_option(k, v)
        p = tg.TgHipPassDesc(0, a.spp, 1234, 0, 1, 0)
        ssum = np.empty((h, w, 3), np.float32)
        cnt = np.empty((h, w), np.uint32)
        for rc in (tg.lib.tghip_clear_framebuffer(ctx), tg.lib.tghip_render_pass(ctx, C.byref(p)), tg.lib.tghip_wait(ctx),
                   tg.lib.tghip_download_framebuffer(ctx, ssum.ctypes.data, cnt.ctypes.data, w*h)):
            if rc != 0:
                raise SystemExit(tg.lib.tghip_last_error(ctx).decode())
        return ssum/a.spp

    ref = None
    print("instances10k %dx%d, %d spp; pixels that differ from the two-level BVH2 shadow walk:" % (w, h, a.spp))
    for name, opts in (("BVH2 shadow walk", dict(wide_shadow=0)),
                       ("wide walk, with the join (as shipped)", dict(wide_shadow=1, inst_shadow_join=1)),
                       ("wide walk, counting its visits", dict(count_traversal=1)),
                       ("wide walk WITHOUT the join", dict(count_traversal=0, inst_shadow_join=0)),
                       ("wide walk WITHOUT the join, again", dict()),
                       ("wide walk, with the join, again", dict(inst_shadow_join=1))):
        m = render(**opts)
        if ref is None:
            ref = m
        d = np.abs(m - ref).max(axis=-1)
        print("  %-40s image mean %.6f   differing pixels %7.3f %%   max |diff| %.4g" % (name, float(m.mean()), 100.0*float((d > 0).mean()), float(d.max())))
    r.close()


if __name__ == "__main__":
    main()
