"""How many different values a median tile's footprint holds, per link of Extractor::foreground's chain (what decides rank form / full form in k_median_cols)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from poppy_amd import capi, synth
W, H = 1920, 1080
Cw, R = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 32
c = capi.Context(0)
for name, bgr in (("photo", synth.photo_pair(W, H)[0]), ("photo2", synth.photo_pair(W, H)[1]), ("shapes", synth.gen(W, H, 1234)), ("texture", synth.textured_bgr(W, H, 7))):
    g = np.ascontiguousarray(bgr[:, :, 1])
    gg = (bgr[:, :, 0].astype(np.float32) * 0.114 + bgr[:, :, 1] * 0.587 + bgr[:, :, 2] * 0.299 + 0.5).astype(np.uint8)
    cur = gg
    for i in range(1, 12):
        k = 8 * i + 1; r = k // 2
        counts = []
        for y0 in range(0, H, R):
            for x0 in range(0, W, Cw):
                fp = cur[max(y0 - r, 0): min(y0 + R + r, H), max(x0 - r, 0): min(x0 + Cw + r, W)]
                counts.append(len(np.unique(fp)))
        counts = np.array(counts)
        print(f"{name} link {i} ksize {k}: tiles {len(counts)}, distinct values median {int(np.median(counts))}, max {counts.max()}; <=64: {np.mean(counts <= 64):.2f}, <=96: {np.mean(counts <= 96):.2f}, <=128: {np.mean(counts <= 128):.2f}", flush=True)
        cur = c.median_blur(cur, k, 1)
