#!/usr/bin/env python3
"""Library vs oracle on non-synthetic content, every set-up product and three chained frames, with the differences counted (not asserted)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from poppy_amd import capi, synth
for name, (a, b) in (("photo 320x180", synth.photo_pair(320, 180)), ("textured 256x192", (synth.textured_bgr(256, 192, 7), synth.textured_bgr(256, 192, 8)))):
    want = O.pair_setup(a, b)
    for direct in (False, True):
        c = capi.Context(0, number_of_frames=3)
        c.set_gabor_direct(direct)
        nf, det = c.pair_begin(a, b)
        p1, p2 = c.pair_points()
        g = c.fetch("gabor2")
        gd = (g.view(np.uint32) != want["gabor2"].view(np.uint32))
        frames = c.morph_frames(-1.0)
        ref = O.morph(a, b, 3, setup=want)
        same_pts = p1.shape == want["points1"].shape and np.array_equal(p1, want["points1"]) and np.array_equal(p2, want["points2"])
        print(f"{name} gabor {'direct' if direct else 'fft'}: nfeatures {nf} vs {want['nfeatures']}, detail equal {det == want['detail']}, points equal {same_pts} ({len(p1)} vs {len(want['points1'])}), "
              f"gabor2 differing floats {int(gd.sum())} (max abs {float(np.abs(g - want['gabor2']).max()):.3g}), frames differing bytes {[int((x != y).sum()) for x, y in zip(frames, ref)]}")
        c.close()
