#!/usr/bin/env python3
"""How long hipMalloc / hipFree / hipHostMalloc take by size on this box (the fixed costs of a cold context)."""
import ctypes as C
import time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
p = C.c_void_p()
t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), 1 << 20); hip.hipDeviceSynchronize(); print("first call (runtime initialisation) %.3f s" % (time.perf_counter() - t0)); hip.hipFree(p)
for gb in (0.25, 1, 4, 10, 16):
    n = int(gb * (1 << 30))
    t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); t1 = time.perf_counter()
    hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipFree(p); t3 = time.perf_counter()
    print("hipMalloc %5.2f GB: %.3f s (rc %d), first memset %.3f s, hipFree %.3f s" % (gb, t1 - t0, rc, t2 - t1, t3 - t2), flush=True)
for gb in (0.25, 1, 3.2):
    n = int(gb * (1 << 30))
    t0 = time.perf_counter(); rc = hip.hipHostMalloc(C.byref(p), n, 0); t1 = time.perf_counter()
    hip.hipHostFree(p); t2 = time.perf_counter()
    print("hipHostMalloc %5.2f GB: %.3f s (rc %d), hipHostFree %.3f s" % (gb, t1 - t0, rc, t2 - t1), flush=True)
