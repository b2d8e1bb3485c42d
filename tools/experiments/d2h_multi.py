#!/usr/bin/env python3
"""Device-to-pinned-host frame copies from N threads / streams at queue depth D each (the pool's writer path: a context keeps at most ring-size copies in flight and
waits for them in order): aggregate GB/s.  usage: d2h_multi.py"""
import ctypes, threading, time
hip = ctypes.CDLL("libamdhip64.so")
n = 1920 * 1080 * 3
def mk(depth):
    d = ctypes.c_void_p(); s = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(n)) == 0
    hs, evs = [], []
    for _ in range(depth):
        h = ctypes.c_void_p(); assert hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(n), 0) == 0; hs.append(h)
        e = ctypes.c_void_p(); assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0; evs.append(e)
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return d, hs, evs, s
def worker(x, reps, depth):
    d, hs, evs, s = x
    for i in range(reps):
        k = i % depth
        if i >= depth: hip.hipEventSynchronize(evs[k])            # the ring slot's previous copy has landed (the writer took it)
        hip.hipMemcpyAsync(hs[k], d, ctypes.c_size_t(n), 2, s)
        hip.hipEventRecord(evs[k], s)
    hip.hipStreamSynchronize(s)
for threads in (1, 2, 3, 6):
    for depth in (1, 2, 3):
        xs = [mk(depth) for _ in range(threads)]
        for x in xs: worker(x, 5, depth)
        R = 150
        th = [threading.Thread(target=worker, args=(x, R, depth)) for x in xs]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print(f"{threads} threads x depth {depth}: {threads * R * n / dt / 1e9:6.1f} GB/s  = {threads * R / dt:7.0f} frames/s")
