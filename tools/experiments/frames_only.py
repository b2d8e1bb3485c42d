#!/usr/bin/env python3
"""Frame loop only, for kernel traces: frames_only.py WIDTH HEIGHT N MODE(chain|phase) REPS   (rocprofv3 --kernel-trace --stats -- python3 ...)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
w, h, n, mode, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
a, b = synth.gen_pair(w, h, seed=1234)
c = capi.Context(0, number_of_frames=n)
c.pair_begin(a, b)
ts = np.array([capi.lib().poppy_frame_ratio(j, n, -1.0) for j in range(n)]) if mode == "chain" else np.arange(1, n + 1) / float(n + 1)
c.reset(); c.render_many(ts, chain=(mode == "chain")); c.sync()
t0 = time.perf_counter()
for _ in range(reps):
    c.reset(); c.render_many(ts, chain=(mode == "chain"))
c.sync()
dt = time.perf_counter() - t0
print(f"{w}x{h} {mode}: {reps * n / dt:.1f} frames/s, {dt / (reps * n) * 1e6:.1f} us per frame, warp kernels {c.warp_counts()}")
