#!/usr/bin/env python3
"""Cost of the host-side HIP calls on the one-shot path (64 MiB = 2^20 affine points): hipHostRegister / Unregister, hipMalloc / hipFree, pinned H2D."""
import ctypes as C, time, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
n = 64 << 20
buf = np.ones(n, dtype=np.uint8)
d = C.c_void_p()
def T(name, fn, reps=5):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
        assert r in (0, None), (name, r)
    print("%-40s min %.3f ms  median %.3f ms" % (name, 1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2]))
hip.hipSetDevice(0)
hip.hipMalloc(C.byref(d), C.c_size_t(n)); hip.hipFree(d)
def mal():
    r = hip.hipMalloc(C.byref(d), C.c_size_t(n)); return r
def fre():
    return hip.hipFree(d)
for _ in range(3):
    T("hipMalloc 64 MiB", mal, 1); T("hipFree", fre, 1)
hip.hipMalloc(C.byref(d), C.c_size_t(n))
p = C.c_void_p(buf.ctypes.data)
for _ in range(3):
    T("hipHostRegister 64 MiB", lambda: hip.hipHostRegister(p, C.c_size_t(n), 0), 1)
    T("hipMemcpy H2D pinned 64 MiB", lambda: hip.hipMemcpy(d, p, C.c_size_t(n), 1), 3)
    T("hipHostUnregister", lambda: hip.hipHostUnregister(p), 1)
T("hipMemcpy H2D pageable 64 MiB", lambda: hip.hipMemcpy(d, p, C.c_size_t(n), 1), 3)
