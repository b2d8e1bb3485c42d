"""Times the median kernels alone: every window of Extractor::foreground's chain in each form of poppy_hip_median_blur, on three kinds of
content.  Run under rocprofv3 --kernel-trace --stats; tools/rocprof_summary.py gives the per-kernel averages.
usage: median_forms.py [W H] [forms, e.g. 123] [reps]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from poppy_amd import capi, synth
W = int(sys.argv[1]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
forms = [int(ch) for ch in (sys.argv[3] if len(sys.argv) > 3 else "123")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
c = capi.Context(0)
content = {"shapes": synth.gen(W, H, 1234), "photo": synth.photo_pair(W, H)[0], "texture": synth.textured_bgr(W, H, 7)}
for name, bgr in content.items():
    cur = np.ascontiguousarray(bgr[:, :, 1])
    for i in range(1, 12):
        k = 8 * i + 1
        outs = []
        for f in forms:
            for _ in range(reps):
                out = c.median_blur(cur, k, f)
            outs.append(out)
        ok = all(np.array_equal(outs[0], o) for o in outs[1:])
        print(f"{name} link {i} ksize {k}: forms {forms} agree: {ok}; distinct values in the source {len(np.unique(cur))}", flush=True)
        cur = outs[0]
