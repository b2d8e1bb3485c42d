#!/usr/bin/env python3
"""Cost of the pair set-up with and without auto-align at 1080p (synthetic pair), and of one standalone warpAffine."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poppy_amd import capi, synth

W, H = 1920, 1080
a, b = synth.gen_pair(W, H)
for flag in (0, 1):
    c = capi.Context(0, enable_auto_align=flag)
    c.pair_begin(a, b)
    t0 = time.perf_counter()
    for _ in range(3):
        nf, _d = c.pair_begin(a, b)
    dt = (time.perf_counter() - t0) / 3
    print(f"pair_begin enable_auto_align={flag}: {dt * 1e3:.1f} ms, nfeatures {nf}, {len(c.pair_points()[0])} point pairs")
    c.close()
c = capi.Context(0)
M = [0.98, 0.07, 3.25, -0.05, 1.03, -2.5]
c.warp_affine(b, M)
t0 = time.perf_counter()
for _ in range(10):
    c.warp_affine(b, M)
print(f"warp_affine host->host 1080p: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms (includes both PCIe copies and two allocations)")
