#!/usr/bin/env python3
"""Pair set-up time (host images in, poppy_hip_pair_begin) at 1080p and 4K: setup_q.py [torch]   (GPU_MAX_HW_QUEUES from the environment)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch; torch.zeros(1, device="cuda")
from poppy_amd import capi, synth
out = []
for (w, h) in [(1920, 1080), (3840, 2160)]:
    a, b = synth.gen_pair(w, h)
    c = capi.Context(0, number_of_frames=60)
    c.pair_begin(a, b); c.pair_begin(a, b)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); c.pair_begin(a, b); ts.append(time.perf_counter() - t0)
    out.append(f"{w}x{h} {min(ts) * 1e3:.2f} ms best, {sum(ts) / len(ts) * 1e3:.2f} mean")
print(f"queues {os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: " + "; ".join(out))
