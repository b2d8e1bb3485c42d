#!/usr/bin/env python3
"""K resident pairs on K contexts of ONE process, one host thread each, 60 chained 1080p frames per sequence: aggregate frames/s with the frames left in HBM and with
every frame handed to a writer.  (No pair set-up inside: what the contexts of a pool could deliver if set-ups were free.)   usage: concurrent_resident.py K [steps]"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
K = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = np.array([capi.lib().poppy_frame_ratio(j, 60, -1.0) for j in range(60)])
ctxs = []
for k in range(K):
    a, b = synth.gen_pair(1920, 1080, seed=1234 + k)
    c = capi.Context(0, number_of_frames=60); c.pair_begin(a, b); ctxs.append(c)
for writer in (False, True):
    def work(c, n):
        for _ in range(n):
            c.reset()
            if writer: c.render_many_counted(shapes, chain=True)
            else: c.render_many(shapes, chain=True)
        c.sync()
    for c in ctxs: work(c, 1)
    th = [threading.Thread(target=work, args=(c, steps)) for c in ctxs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{K} resident pairs on {K} contexts of one process, {'with the writer' if writer else 'frames left in HBM'}: {K * steps * 60 / dt:.0f} frames/s")
