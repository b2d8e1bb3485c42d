#!/usr/bin/env python3
"""The two forms of the Gabor banks (tiled FFTs + the hand-over of doubtful pixels; direct double sums) on full-size images of three kinds of content:
every float of both banks' outputs must be the same bits.   usage: gabor_forms_big.py [W H]..."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
sizes = [(1920, 1080), (3840, 2160), (1277, 719)]
bad = 0
for w, h in sizes:
    for kind in ("photo", "synthetic", "textured"):
        a, b = {"synthetic": lambda: synth.gen_pair(w, h, seed=77), "textured": lambda: (synth.textured_bgr(w, h, 17), synth.textured_bgr(w, h, 18)),
                "photo": lambda: synth.photo_pair(w, h)}[kind]()
        for img in (a, b):
            gf = np.ascontiguousarray(img[:, :, 1])
            c = capi.Context(0, number_of_frames=1)
            capi.gabor_doubt()
            c.set_gabor_direct(False); x, fx = c.orb_input(gf), c.gabor_field(img)
            doubt = capi.gabor_doubt()
            c.set_gabor_direct(True); y, fy = c.orb_input(gf), c.gabor_field(img)
            c.close()
            n31 = int((x["gb"].view(np.uint32) != y["gb"].view(np.uint32)).sum()); n13 = int((fx.view(np.uint32) != fy.view(np.uint32)).sum())
            ng = int((x["g"] != y["g"]).sum())
            bad += (n31 + n13 + ng) != 0
            print(f"{w}x{h} {kind}: differing floats 31x31 bank {n31}, 13x13 bank {n13}, ORB input bytes {ng}; doubtful (near zero, near midpoint, pixels redone) {doubt}", flush=True)
print("MISMATCH" if bad else "all equal")
