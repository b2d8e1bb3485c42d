#!/usr/bin/env python3
"""Which parts of a pair's work overlap when two contexts of one GPU run them from two host threads?
usage: overlap_probe.py [--torch]   (--torch imports torch first, as bench.py does: its bundled HIP runtime then serves the library)"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--torch" in sys.argv:
    import torch
    torch.cuda.init()
import numpy as np
from poppy_amd import capi, synth

W, H, N = 1920, 1080, 60
pairs = [synth.gen_pair(W, H, seed=1234 + k) for k in range(2)]
ctx = [capi.Context(0, number_of_frames=N) for _ in range(2)]
shapes = np.array([capi.lib().poppy_frame_ratio(j, N, -1.0) for j in range(N)])


def setup(i, reps):
    for _ in range(reps):
        ctx[i].pair_begin(*pairs[i])


def frames(i, reps, writer):
    for _ in range(reps):
        ctx[i].reset()
        if writer:
            ctx[i].render_many_counted(shapes, chain=True)
        else:
            ctx[i].render_many(shapes, chain=True)
    ctx[i].sync()


def timed(jobs):
    th = [threading.Thread(target=f, args=a) for f, a in jobs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) * 1e3


for i in range(2):
    setup(i, 2); frames(i, 1, True)
R = 10
print("setup x%d alone            : %.1f ms each" % (R, timed([(setup, (0, R))]) / R))
print("setup || setup             : %.1f ms per pair of set-ups" % (timed([(setup, (0, R)), (setup, (1, R))]) / R))
print("frames alone (HBM)         : %.1f ms per 60" % (timed([(frames, (0, R, False))]) / R))
print("frames || frames (HBM)     : %.1f ms per 2x60" % (timed([(frames, (0, R, False)), (frames, (1, R, False))]) / R))
print("frames alone (writer)      : %.1f ms per 60" % (timed([(frames, (0, R, True))]) / R))
print("frames || frames (writer)  : %.1f ms per 2x60" % (timed([(frames, (0, R, True)), (frames, (1, R, True))]) / R))
print("setup || frames (HBM)      : %.1f ms per (set-up, 60)" % (timed([(setup, (0, R)), (frames, (1, R, False))]) / R))
print("setup || frames (writer)   : %.1f ms per (set-up, 60)" % (timed([(setup, (0, R)), (frames, (1, R, True))]) / R))
# the same loops on the MAIN thread (no threading.Thread around them)
t0 = time.perf_counter(); frames(0, R, True); print("frames alone (writer), main thread : %.1f ms per 60" % ((time.perf_counter() - t0) * 1e3 / R))
t0 = time.perf_counter(); frames(0, R, False); print("frames alone (HBM), main thread    : %.1f ms per 60" % ((time.perf_counter() - t0) * 1e3 / R))
t0 = time.perf_counter(); frames(1, R, True); print("frames alone (writer), main thread, ctx 1 : %.1f ms per 60" % ((time.perf_counter() - t0) * 1e3 / R))
