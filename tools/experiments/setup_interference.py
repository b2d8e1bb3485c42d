#!/usr/bin/env python3
"""How much do pair set-ups and chained frames cost each other on one GPU?  R contexts render resident pairs (60 chained 1080p frames per sequence, every frame to a writer)
while S contexts run pair set-ups back to back, for `secs` seconds; prints frames/s and set-ups/s.   usage: setup_interference.py R S [secs]"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
R, S = int(sys.argv[1]), int(sys.argv[2]); secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
shapes = np.array([capi.lib().poppy_frame_ratio(j, 60, -1.0) for j in range(60)])
pairs = [synth.gen_pair(1920, 1080, seed=1234 + k) for k in range(max(R + S, 1))]
rc = []; sc = []
for k in range(R):
    c = capi.Context(0, number_of_frames=60); c.pair_begin(*pairs[k]); c.reset(); c.render_many_counted(shapes, chain=True); rc.append(c)
for k in range(S):
    c = capi.Context(0, number_of_frames=60); c.pair_begin(*pairs[R + k]); sc.append(c)
stop = False; frames = [0] * R; setups = [0] * S
def render(i):
    while not stop:
        rc[i].reset(); frames[i] += rc[i].render_many_counted(shapes, chain=True)
def setup(i):
    a, b = pairs[R + i]
    while not stop:
        sc[i].pair_begin(a, b); setups[i] += 1
th = [threading.Thread(target=render, args=(i,)) for i in range(R)] + [threading.Thread(target=setup, args=(i,)) for i in range(S)]
t0 = time.perf_counter()
for t in th: t.start()
time.sleep(secs); stop = True
for t in th: t.join()
dt = time.perf_counter() - t0
print(f"{R} rendering + {S} setting up: {sum(frames) / dt:.0f} frames/s, {sum(setups) / dt:.1f} set-ups/s ({dt / max(sum(setups), 1) * 1e3 * max(S, 1):.2f} ms per set-up per context)")
