import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from poppy_amd import capi, synth
for (w,h) in [(1920,1080)]:
    a,b = synth.gen_pair(w,h)
    c = capi.Context(0, number_of_frames=60)
    c.pair_begin(a,b)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); nf,d = c.pair_begin(a,b); ts.append(time.perf_counter() - t0)
    p1,p2=c.pair_points()
    print(f"{w}x{h}: pair_begin {np.median(ts)*1e3:.2f} ms median of {n}, min {min(ts)*1e3:.2f} (nfeatures {nf}, {len(p1)} point pairs)")
    t0=time.perf_counter(); fr=c.morph_frames(-1.0); dt=time.perf_counter()-t0
    print(f"  60 chained frames incl. download: {dt*1e3:.1f} ms")
