#!/usr/bin/env python3
"""Frame hand-off rate (chained frames -> pinned host ring -> writer) with one context, then with a second context alive, and the raw
device-to-pinned-host copy rate of this box."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth

W, H, N = 1920, 1080, 60
a, b = synth.gen_pair(W, H, seed=1234)
shapes = np.array([capi.lib().poppy_frame_ratio(j, N, -1.0) for j in range(N)])


def rate(c, writer, reps=10):
    c.reset(); (c.render_many_counted if writer else c.render_many)(shapes, chain=True); c.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.reset(); (c.render_many_counted if writer else c.render_many)(shapes, chain=True)
    c.sync()
    return (time.perf_counter() - t0) * 1e3 / reps


c0 = capi.Context(0, number_of_frames=N)
c0.pair_begin(a, b)
print("one context : %.1f ms per 60 (HBM), %.1f ms per 60 (writer)" % (rate(c0, False), rate(c0, True)))
c1 = capi.Context(0, number_of_frames=N)
c1.pair_begin(a, b)
print("two alive   : %.1f ms per 60 (HBM), %.1f ms per 60 (writer) on the first; %.1f (writer) on the second" % (rate(c0, False), rate(c0, True), rate(c1, True)))
c1.close()
print("second gone : %.1f ms per 60 (writer)" % rate(c0, True))
hip = ctypes.CDLL("libamdhip64.so")
n = W * H * 3
d = ctypes.c_void_p(); h = ctypes.c_void_p(); s = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(n)); hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(n), 0)
hip.hipStreamCreateWithFlags(ctypes.byref(s), 1)
for _ in range(3):
    hip.hipMemcpyAsync(h, d, ctypes.c_size_t(n), 2, s)
hip.hipStreamSynchronize(s)
t0 = time.perf_counter()
for _ in range(100):
    hip.hipMemcpyAsync(h, d, ctypes.c_size_t(n), 2, s)
hip.hipStreamSynchronize(s)
dt = time.perf_counter() - t0
print("raw D2H     : %.1f GB/s (%.0f us per 1080p frame)" % (100 * n / dt / 1e9, dt / 100 * 1e6))
