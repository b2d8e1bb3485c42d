#!/usr/bin/env python3
"""A/B of the fused warp kernel's forms on ONE box (poppy_hip_set_warp_variant): per size, every variant renders the same phase frame of the
synthetic pair — trImg1 / trImg2 / frame compared byte for byte with variant 0 — and is then relaunched back to back (poppy_hip_time_last_warp).
usage: warp_variants.py [variants, hex, comma separated] [reps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
variants = [int(v, 16) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,34,24,15").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for w, h in ((1920, 1080), (3840, 2160)):
    a, b = synth.gen_pair(w, h, seed=1234)
    c = capi.Context(0, number_of_frames=60)
    c.pair_begin(a, b)
    ref = None
    for v in variants:
        capi.Context.set_warp_variant(v)
        outs = []
        for t in (0.5, 0.13):
            frame = c.render(t, t)
            outs.append((frame.copy(), c.fetch("trImg1"), c.fetch("trImg2")))
        if ref is None:
            ref = outs
        same = all(np.array_equal(x, y) for o, r in zip(outs, ref) for x, y in zip(o, r))
        nd = sum(int((x != y).sum()) for o, r in zip(outs, ref) for x, y in zip(o, r))
        times = [float("nan")] * 5
        for t in (0.5, 0.13, 0.02):                        # a frame that takes the fused kernel (a sliver triangle sends a frame to the general one)
            c.render(t, t)
            if c.last_warp_kind() == 2:
                times = sorted(c.time_last_warp(reps) * 1e3 for _ in range(5))
                break
        print(f"{w}x{h} variant {v:#04x}: {'identical' if same else 'DIFFERENT (%d bytes)' % nd}  relaunch {times[0]:.2f} / {times[2]:.2f} / {times[-1]:.2f} us (min / median / max of 5 x {reps}), kinds {c.warp_counts()}", flush=True)
    capi.Context.set_warp_variant(-1)
    del c
