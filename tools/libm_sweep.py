"""Every float32 bit pattern through csrc/hip/pt_libm.h (compiled for the host: oracle/libm_host.so) against the host libm.
TEST INFRASTRUCTURE.  About 15 s on 32 cores; prints the number of mismatches per function (all zero on the image's glibc 2.35)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "oracle", "libm_host.so"))
lib.libm_host_sweep.restype = C.c_ulonglong
lib.libm_host_sweep.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint]
for fn, name in enumerate(("sinf", "cosf", "logf", "expf", "sincos: sin", "sincos: cos")):
    print("%-12s mismatches: %d (x >= 0), %d (x < 0)" % (name, lib.libm_host_sweep(fn, 0, 0x7F800000, 1), lib.libm_host_sweep(fn, 0x80000000, 0xFF800000, 1)))
