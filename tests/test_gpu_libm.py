"""The device's libm (csrc/hip/pt_libm.h, pt_math.h: sinfH / cosfH / sincosfH / logfH / expfH / atan2fH / powfH / cbrtfH / acosfExact -- glibc's algorithms restated)
evaluated ON THE DEVICE through tghip_debug_libm against the host libm, bit for bit, on the arguments the kernels produce: angles
2 pi xi and pi v, 1 - xi for the logarithm, negative optical depths for the exponential, cosines for the arc cosine."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
import tungsten_amd as tg
from tungsten_amd import capi

pytestmark = pytest.mark.gpu


def _host():
    lib = C.CDLL(os.path.join(scenes.ROOT, "oracle", "libm_host.so"))
    lib.libm_host_ref.restype = None
    lib.libm_host_ref.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    return lib


def test_device_libm_is_the_host_libm_bit_for_bit(tmp_path):
    if "fma" not in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        pytest.skip("host CPU without FMA3: glibc runs its non-FMA variants here")
    host = _host()
    r = tg.Renderer(scenes.cornell(tmp_path, resolution=(16, 16), spp=1))
    rng = np.random.default_rng(11)
    n = 1 << 21
    xi = rng.random(n, dtype=np.float32)
    grid = (np.arange(n, dtype=np.float32) + 0.5)/n            # every 2^-21 of [0, 1): the quadrant boundaries of 2 pi xi
    two_pi, pi = np.float32(3.1415926536*2), np.float32(3.1415926536)
    cases = [
        (capi.TGHIP_LIBM_SINF, [xi*two_pi, grid*two_pi, xi*pi, (xi - np.float32(0.5))*two_pi, xi*np.float32(1e-3), xi*np.float32(119.0), -xi*np.float32(119.0)]),
        (capi.TGHIP_LIBM_COSF, [xi*two_pi, grid*two_pi, xi*pi, (xi - np.float32(0.5))*two_pi, xi*np.float32(1e-3), xi*np.float32(119.0), -xi*np.float32(119.0)]),
        (capi.TGHIP_LIBM_SINCOS_SIN, [xi*two_pi, grid*two_pi, xi*pi, -xi*two_pi]),
        (capi.TGHIP_LIBM_SINCOS_COS, [xi*two_pi, grid*two_pi, xi*pi, -xi*two_pi]),
        (capi.TGHIP_LIBM_LOGF, [np.float32(1.0) - xi, np.float32(1.0) - grid, xi + np.float32(1e-30), xi*np.float32(1e30) + np.float32(1e-30), np.float32(1.0) + xi*np.float32(50.0)]),
        (capi.TGHIP_LIBM_EXPF, [-xi*np.float32(87.9), xi*np.float32(87.9), -xi, -xi*np.float32(1e-4), -grid*np.float32(20.0)]),
        (capi.TGHIP_LIBM_ACOSF, [xi*np.float32(2.0) - np.float32(1.0), np.float32(1.0) - xi*np.float32(1e-4), grid*np.float32(2.0) - np.float32(1.0)]),
    ]
    for fn, arrays in cases:
        for x in arrays:
            x = np.ascontiguousarray(x, np.float32)
            got = r.debug_libm(fn, x)
            want = np.empty_like(x)
            host.libm_host_ref(fn, x.ctypes.data, want.ctypes.data, x.size)
            bad = got.view(np.uint32) != want.view(np.uint32)
            assert not bad.any(), "fn %d: %d of %d differ, first x = %r: device %r, host %r" % (fn, int(bad.sum()), x.size, x[bad][0], got[bad][0], want[bad][0])
    # atan2f / powf / cbrtf (round 4: called by the kernels): the operands the path produces -- direction components for the texture coordinates of
    # environment maps and spheres (all four quadrants, axes, tiny and zero components), 1 + tau/p and optical depths for the Davis
    # transmittances, z + 1/z of the Rayleigh phase function -- and arbitrary bit patterns.  (oracle/libm_host.cpp numbers the host's
    # functions differently: 9 = atan2f and 10 = powf over interleaved pairs, 8 = cbrtf.)
    def pairs(a, b):
        return np.ascontiguousarray(np.stack([a, b], axis=1), np.float32).reshape(-1)
    sgn = np.where(rng.random(n) < 0.5, np.float32(-1.0), np.float32(1.0))
    d1, d2 = (xi*np.float32(2.0) - np.float32(1.0)), (grid*np.float32(2.0) - np.float32(1.0))
    zeros = np.zeros(n, np.float32)
    bits = rng.integers(0, 1 << 32, 2*n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    two_arg = [
        (capi.TGHIP_LIBM_ATAN2F, 9, [pairs(d1, d2), pairs(d2, d1*sgn), pairs(d1*np.float32(1e-4), d2), pairs(d1, d2*np.float32(1e-6)), pairs(zeros, d2), pairs(d1, zeros),
                                     pairs(-zeros, d2), pairs(d1, -zeros), pairs(d1, np.ones(n, np.float32)), bits]),
        (capi.TGHIP_LIBM_POWF, 10, [pairs(np.float32(1.0) + xi*np.float32(8.0), -(np.float32(0.5) + grid*np.float32(6.0))), pairs(xi*np.float32(20.0) + np.float32(1e-6), np.float32(1.0) - grid*np.float32(0.9)),
                                    pairs(np.float32(1.0) - xi*np.float32(0.999), -np.float32(1.0)/(np.float32(0.1) + grid*np.float32(5.0))), pairs(xi + np.float32(0.5), d2*np.float32(60.0))]),
    ]
    for fn, host_fn, arrays in two_arg:
        for x in arrays:
            got = r.debug_libm(fn, x)
            want = np.empty(x.size//2, np.float32)
            host.libm_host_ref(host_fn, x.ctypes.data, want.ctypes.data, want.size)
            bad = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
            assert not bad.any(), "fn %d: %d of %d differ, first (%r, %r): device %r, host %r" % (
                fn, int(bad.sum()), want.size, x[0::2][bad][0], x[1::2][bad][0], got[bad][0], want[bad][0])
    for x in (xi + np.float32(1.0)/np.maximum(xi, np.float32(1e-3)), d1*np.float32(30.0), xi*np.float32(1e-30), bits[:n]):
        x = np.ascontiguousarray(x, np.float32)
        got = r.debug_libm(capi.TGHIP_LIBM_CBRTF, x)
        want = np.empty_like(x)
        host.libm_host_ref(8, x.ctypes.data, want.ctypes.data, x.size)
        bad = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        assert not bad.any(), "cbrtf: %d of %d differ, first x = %r: device %r, host %r" % (int(bad.sum()), x.size, x[bad][0], got[bad][0], want[bad][0])
    # tanf (the Oren-Nayar BSDF's tan(beta), tan((alpha + beta)/2): angles in [0, pi/2]; restated for |x| < 120 -- host id 12)
    for x in (xi*np.float32(1.5707964), grid*np.float32(1.5707964), d1*np.float32(3.0), d1*np.float32(100.0), xi*np.float32(1e-5), np.float32(1.5707964) - xi*np.float32(1e-3)):
        x = np.ascontiguousarray(x, np.float32)
        got = r.debug_libm(capi.TGHIP_LIBM_TANF, x)
        want = np.empty_like(x)
        host.libm_host_ref(12, x.ctypes.data, want.ctypes.data, x.size)
        bad = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        assert not bad.any(), "tanf: %d of %d differ, first x = %r: device %r, host %r" % (int(bad.sum()), x.size, x[bad][0], got[bad][0], want[bad][0])
    # Embree's rcp() of its triangle test (pt_scene.h: rcppsIntel / embreeRcp) against the oracle's restatement (oracle.c: intel_rcpps / embree_rcp,
    # itself held to the instruction by tests/test_host.py): every exponent x every table index x low mantissa bits, both signs, random patterns
    import oracle_lib
    e = np.arange(0, 256, dtype=np.uint32)[:, None, None] << 23
    idx = np.arange(0, 2048, dtype=np.uint32)[None, :, None] << 12
    low = np.array([0, 1, 0x7ff, 0xfff], np.uint32)[None, None, :]
    pat = np.concatenate([(e | idx | low).reshape(-1), (e | idx | low).reshape(-1) | np.uint32(0x80000000), rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)])
    x = np.ascontiguousarray(pat.view(np.float32))
    for fn, raw in ((capi.TGHIP_LIBM_RCPPS, 1), (capi.TGHIP_LIBM_EMBREE_RCP, 0)):
        got = r.debug_libm(fn, x)
        want = np.empty_like(x)
        oracle_lib._lib.oracle_embree_rcp(raw, x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
        bad = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        assert not bad.any(), "rcp fn %d: %d of %d differ, first x = %r: device %r, oracle %r" % (fn, int(bad.sum()), x.size, x[bad][0], got[bad][0], want[bad][0])
    # double precision (round 6: AtmosphericMedium::inverseOpticalDepth -- std::erf(s t0), std::exp of s^2 (h - r)(h + r), Erf::erfInv's std::log(q) and std::sqrt):
    # pt_libm.h's expD / logD / erfD and the device's correctly rounded sqrt against the host libm (oracle/libm_host.cpp: libm_host_refd, 0 exp, 1 log, 2 erf, 3 sqrt)
    host.libm_host_refd.restype = None
    host.libm_host_refd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    u = rng.random(n)
    bits64 = rng.integers(0, 1 << 63, n, dtype=np.uint64).view(np.float64)          # arbitrary non-negative bit patterns (subnormals, infinities, NaNs among them)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2e-308, 1.7e308, 709.78, 709.79, -708.4, -745.1, -745.2, 1 - 2.0**-53, 1 + 2.0**-52, 0.9375, 1.0647, 0.5, 0.25, 6.0, -6.0, 0.84375, 1.25, 2.857142857142857], np.float64)
    dcases = [
        (capi.TGHIP_LIBM_EXPD, 0, [u*1465.0 - 750.0, -u*40.0, -u, (u - 0.5)*1e-12, -u*u*30.0 - 0.5625, bits64, -bits64, special]),
        (capi.TGHIP_LIBM_LOGD, 1, [u*0.5, u**4*0.5, 0.9 + u*0.2, u*2.0, u*1e300, u*1e-300, bits64, special]),
        (capi.TGHIP_LIBM_ERFD, 2, [u*8.0 - 4.0, u*14.0 - 7.0, (u - 0.5)*1e-6, (u - 0.5)*1e-300, bits64, -bits64, special]),
        (capi.TGHIP_LIBM_SQRTD, 3, [u, u*60.0, -2.0*np.log(np.maximum(u, 1e-300)), bits64, special]),
    ]
    for fn, host_fn, arrays in dcases:
        for x in arrays:
            x = np.ascontiguousarray(x, np.float64)
            got = r.debug_libm(fn, x)
            want = np.empty_like(x)
            host.libm_host_refd(host_fn, x.ctypes.data, want.ctypes.data, x.size)
            bad = (got.view(np.uint64) != want.view(np.uint64)) & ~(np.isnan(got) & np.isnan(want))
            assert not bad.any(), "double fn %d: %d of %d differ, first x = %r: device %r, host %r" % (fn, int(bad.sum()), x.size, x[bad][0], got[bad][0], want[bad][0])
    # logf / expf are glibc's for EVERY float (pt_libm.h: logfAll / expfAll, all 2^32 bit patterns checked on the host); here the special cases
    # whose results are not subnormal: zeros, negatives, infinities, NaN, the overflow / underflow thresholds
    for fn, x in ((capi.TGHIP_LIBM_LOGF, [0.0, -0.0, -1.0, -1e-30, np.inf, -np.inf, np.nan, 1.0, 3.4e38, 1.2e-38]),
                  (capi.TGHIP_LIBM_EXPF, [0.0, -0.0, 88.0, 88.5, 88.72, 88.73, 100.0, np.inf, -np.inf, np.nan, -87.0, -87.3, -104.0, -200.0])):
        x = np.array(x, np.float32)
        got = r.debug_libm(fn, x)
        want = np.empty_like(x)
        host.libm_host_ref(fn, x.ctypes.data, want.ctypes.data, x.size)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (fn, x[~same], got[~same], want[~same])
    # outside the restated range sin / cos still answer sensibly (no call site gets there): to float accuracy; exp(-inf) = 0
    big = np.array([150.0, -1e4, 3e5], np.float32)
    assert np.allclose(r.debug_libm(capi.TGHIP_LIBM_SINF, big), np.sin(big.astype(np.float64)), atol=2e-2)
    assert (r.debug_libm(capi.TGHIP_LIBM_EXPF, np.array([-np.inf, -200.0], np.float32)) == 0.0).all()
    r.close()
