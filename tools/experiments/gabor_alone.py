#!/usr/bin/env python3
"""The two Gabor banks alone on the GPU (no other stream busy): orb_input (31 x 31 bank, one plane) and gabor_field (13 x 13 bank, three planes), ten times each, for a kernel
trace.   usage: gabor_alone.py [W H] [synthetic|photo|textured]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
kind = sys.argv[3] if len(sys.argv) > 3 else "photo"
a, b = {"synthetic": lambda: synth.gen_pair(w, h, seed=1234), "textured": lambda: (synth.textured_bgr(w, h, 7), synth.textured_bgr(w, h, 8)),
        "photo": lambda: synth.photo_pair(w, h)}[kind]()
gf = np.ascontiguousarray(a[:, :, 1])
c = capi.Context(0, number_of_frames=1)
for _ in range(10):
    c.orb_input(gf)
    c.gabor_field(b)
print("done")
