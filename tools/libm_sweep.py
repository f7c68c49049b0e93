"""Every float32 bit pattern through csrc/hip/pt_libm.h (compiled for the host: oracle/libm_host.so) against the host libm.
TEST INFRASTRUCTURE.  About a minute on 8 cores; prints the number of mismatches per function (all zero on the image's glibc 2.35)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "oracle", "libm_host.so"))
lib.libm_host_sweep.restype = C.c_ulonglong
lib.libm_host_sweep.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint]
lib.libm_host_sweep2.restype = C.c_ulonglong
lib.libm_host_sweep2.argtypes = [C.c_int, C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
for fn, name in ((0, "atan2f"), (1, "powf")):
    tested = C.c_ulonglong(0)
    bad = lib.libm_host_sweep2(fn, 400000000, 1, C.byref(tested))
    print("%-12s mismatches: %d of %d pseudo-random pairs" % (name, bad, tested.value))
for fn, name in ((0, "sinf"), (1, "cosf"), (2, "logf"), (3, "expf"), (4, "sincos: sin"), (5, "sincos: cos"), (7, "atanf"), (8, "cbrtf"), (12, "tanf")):
    print("%-12s mismatches: %d (x >= 0), %d (x < 0)" % (name, lib.libm_host_sweep(fn, 0, 0x7F800000, 1), lib.libm_host_sweep(fn, 0x80000000, 0xFF800000, 1)))
