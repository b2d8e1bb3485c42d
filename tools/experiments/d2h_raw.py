#!/usr/bin/env python3
"""Raw device-to-pinned-host copy rate: one stream, two streams from one thread, two streams from two threads."""
import ctypes, threading, time
hip = ctypes.CDLL("libamdhip64.so")
n = 1920 * 1080 * 3
def mk():
    d = ctypes.c_void_p(); h = ctypes.c_void_p(); s = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(n)) == 0
    import os
    assert hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(n), int(os.environ.get("D2H_HOST_FLAGS", "0"))) == 0
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return d, h, s
A, B = mk(), mk()
def run(x, reps):
    for _ in range(reps):
        hip.hipMemcpyAsync(x[1], x[0], ctypes.c_size_t(n), 2, x[2])
    hip.hipStreamSynchronize(x[2])
run(A, 5); run(B, 5)
R = 200
t0 = time.perf_counter(); run(A, R); dt = time.perf_counter() - t0
print("one stream               : %.1f GB/s" % (R * n / dt / 1e9))
t0 = time.perf_counter()
for _ in range(R):
    hip.hipMemcpyAsync(A[1], A[0], ctypes.c_size_t(n), 2, A[2]); hip.hipMemcpyAsync(B[1], B[0], ctypes.c_size_t(n), 2, B[2])
hip.hipStreamSynchronize(A[2]); hip.hipStreamSynchronize(B[2]); dt = time.perf_counter() - t0
print("two streams, one thread  : %.1f GB/s total" % (2 * R * n / dt / 1e9))
th = [threading.Thread(target=run, args=(x, R)) for x in (A, B)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
print("two streams, two threads : %.1f GB/s total" % (2 * R * n / dt / 1e9))
