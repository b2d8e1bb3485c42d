#!/usr/bin/env python3
"""What does the writer hand-off cost the chained frame loop?  One context, 60 chained 1080p frames per pass.
  writer_gap.py W H REPS MODE [torch]    MODE = resident | writer | resident-phase | writer-phase        (prints frames/s; torch: import torch first)
  writer_gap.py --db results.db   per-frame kernel time, idle time between a frame's kernels and between frames, from a rocprofv3 --kernel-trace database
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--db":
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
    # the render queue: the one k_unsharp_tile runs on
    q = [r[3] for r in rows if "k_unsharp_tile" in r[0]][0]
    ks = [(n.split("(")[0].split("::")[-1], s, e) for n, s, e, qq in rows if qq == q]
    ends = [i for i, k in enumerate(ks) if "k_unsharp_tile" in k[0]]
    ends = ends[len(ends) // 2:]                      # the second half: steady state
    per, busy, inner, outer = [], [], [], []
    for a, b in zip(ends[:-1], ends[1:]):
        fr = ks[a + 1:b + 1]
        per.append(ks[b][2] - ks[a][2])
        busy.append(sum(e - s for _, s, e in fr))
        outer.append(fr[0][1] - ks[a][2])
        inner.append(sum(fr[i + 1][1] - fr[i][2] for i in range(len(fr) - 1)))
    m = lambda v: sum(v) / len(v) / 1e3
    print(f"{len(per)} frames: {m(per):.1f} us per frame = {m(busy):.1f} us of kernels + {m(inner):.1f} us between a frame's kernels + {m(outer):.1f} us between frames; kernels per frame {len(fr)}")
    other = {}
    for n, s, e, qq in rows:
        if qq != q: other[n.split("(")[0]] = other.get(n.split("(")[0], 0) + 1
    print("other queues:", other)
    sys.exit(0)

if len(sys.argv) > 5 and sys.argv[5] == 'torch':       # torch's bundled HIP runtime becomes the process's runtime
    import torch; torch.cuda.init(); torch.zeros(1, device='cuda')
import numpy as np
from poppy_amd import capi, synth
w, h, reps, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
n = 60
a, b = synth.gen_pair(w, h, seed=1234)
c = capi.Context(0, number_of_frames=n)
c.pair_begin(a, b)
ts = np.array([capi.lib().poppy_frame_ratio(j, n, -1.0) for j in range(n)])
chain = not mode.endswith("-phase")
if not chain: ts = np.arange(1, n + 1) / float(n + 1)
run = (lambda: c.render_many_counted(ts, chain=chain)) if mode.startswith("writer") else (lambda: c.render_many(ts, chain=chain))
c.reset(); run(); c.sync()
t0 = time.perf_counter()
for _ in range(reps):
    c.reset(); run()
c.sync()
dt = time.perf_counter() - t0
print(f"{w}x{h} {mode}: {reps * n / dt:.1f} frames/s, {dt / (reps * n) * 1e6:.1f} us per frame")
