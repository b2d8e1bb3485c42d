#!/usr/bin/env python3
"""Pair set-up on content other than the flat synthetic shapes (bench.py: content_sensitivity): stage times (POPPY_SETUP_TIMING=1 in the environment prints the
library's own stage stamps on stderr), for a kernel trace run under rocprofv3 --kernel-trace --stats.   usage: setup_content.py {synthetic|textured|photo} [W H] [reps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "photo"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
a, b = {"synthetic": lambda: synth.gen_pair(w, h, seed=1234), "textured": lambda: (synth.textured_bgr(w, h, 7), synth.textured_bgr(w, h, 8)),
        "photo": lambda: synth.photo_pair(w, h)}[kind]()
c = capi.Context(0, number_of_frames=60)
c.pair_begin(a, b)
print(f"Gabor banks in one set-up (plane values near zero, near a midpoint, pixels formed again as direct sums): {capi.gabor_doubt()}")
t = []
for _ in range(reps):
    t0 = time.perf_counter(); nf, det = c.pair_begin(a, b); t.append((time.perf_counter() - t0) * 1e3)
p1, _ = c.pair_points()
print(f"{kind} {w}x{h}: pair set-up {sorted(t)[len(t) // 2]:.2f} ms (min {min(t):.2f}), nfeatures {nf}, {len(p1)} point pairs")
