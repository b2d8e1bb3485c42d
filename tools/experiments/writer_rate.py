#!/usr/bin/env python3
"""Chained 1080p frames of a resident pair on one context, with and without the writer hand-off (frames/s)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
a, b = synth.gen_pair(1920, 1080, seed=1234)
c = capi.Context(0, number_of_frames=60)
c.pair_begin(a, b)
ts = np.array([capi.lib().poppy_frame_ratio(j, 60, -1.0) for j in range(60)])
out = []
for counted in (False, True):
    f = (lambda: (c.reset(), c.render_many_counted(ts, chain=True))) if counted else (lambda: (c.reset(), c.render_many(ts, chain=True)))
    f(); c.sync()
    t0 = time.perf_counter()
    for _ in range(40): f()
    c.sync()
    out.append(40 * 60 / (time.perf_counter() - t0))
print(f"resident chained: {out[0]:.0f} frames/s in HBM, {out[1]:.0f} with the writer")
