"""One median kernel, a few launches (for counter passes): median_one.py W H ksize form content reps"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from poppy_amd import capi, synth
W, H, k, form = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
content = sys.argv[5] if len(sys.argv) > 5 else "photo"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
bgr = {"shapes": lambda: synth.gen(W, H, 1234), "photo": lambda: synth.photo_pair(W, H)[0], "texture": lambda: synth.textured_bgr(W, H, 7)}[content]()
c = capi.Context(0)
g = np.ascontiguousarray(bgr[:, :, 1])
for _ in range(reps):
    out = c.median_blur(g, k, form)
print(W, H, k, form, content, int(out.sum()))
