#!/usr/bin/env python3
"""Back-to-back pinned device-to-host copies of one 1080p frame (6.2 MB) for `secs` seconds; prints GB/s.  Run beside frames_only.py processes to see whether the
copy path itself slows down under compute load (tools/experiments/d2h_under_load.sh)."""
import ctypes, sys, time
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
hip = ctypes.CDLL("libamdhip64.so")
n = 1920 * 1080 * 3
d = ctypes.c_void_p(); h = ctypes.c_void_p(); s = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(n)) == 0
assert hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(n), 0) == 0
assert hip.hipStreamCreate(ctypes.byref(s)) == 0
for _ in range(10):
    hip.hipMemcpyAsync(h, d, ctypes.c_size_t(n), 2, s)
hip.hipStreamSynchronize(s)
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        hip.hipMemcpyAsync(h, d, ctypes.c_size_t(n), 2, s)
    hip.hipStreamSynchronize(s); k += 20
dt = time.perf_counter() - t0
print(f"D2H: {k * n / dt / 1e9:.1f} GB/s ({k / dt:.0f} frames/s)")
