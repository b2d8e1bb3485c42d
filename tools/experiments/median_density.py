#!/usr/bin/env python3
"""How sparse the medians' row-step updates are on the chain's own images (1080p synthetic pair): for every median stage the share of
(pixel, window column) pairs whose entering value differs from the leaving one, the share of (wave of 64 columns, byte position) slots in
which ANY lane differs (= atomics issued now), and the warm-up's share of a segment's updates with the launch geometry in use."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from poppy_amd import capi, synth
w, h = 1920, 1080
a, b = synth.gen_pair(w, h)
c = capi.Context(0)
st = c.foreground(a, debug=True)
prev = st["grey"]
for i in range(1, 12):
    k = 8 * i + 1; r = k // 2
    src = prev                                                   # median i reads the previous median's output (grey for the first)
    ys = np.arange(0, h - 1, 7)                                  # a sample of row steps y -> y + 1
    add = src[np.clip(ys + r + 1, 0, h - 1)]; sub = src[np.clip(ys - r, 0, h - 1)]
    diff = add != sub                                            # per (row step, column): the pair entering / leaving differs
    pad = np.pad(diff, ((0, 0), (r, r)), mode="edge")
    # per output column x the window covers columns x - r .. x + r: density = mean over window of diff
    cs = np.cumsum(np.pad(pad, ((0, 0), (1, 0))), axis=1)
    per_lane = (cs[:, k:] - cs[:, :-k]) / k                      # share of differing pairs in each lane's window
    dens = per_lane.mean()
    # wave-level: for byte position dx the lanes x0 .. x0 + 63 read columns x0 + dx .. x0 + 63 + dx: any differing among 64 consecutive columns
    any64 = (np.lib.stride_tricks.sliding_window_view(pad, 64, axis=1).any(axis=2)).mean()
    segs = max(1, 1024 // ((w + 63) // 64)); rows = max((h + segs - 1) // segs, min(h, (k + 1) // 2))
    warm = k * k; steps_now = rows * 2 * k * any64; steps_dense = rows * 2 * k * dens
    print(f"ksize {k:2d}: differing pairs {100*dens:5.1f} %, wave slots with any difference {100*any64:5.1f} %; per lane and segment ({rows} rows): warm-up {warm} adds, "
          f"steps {steps_now:7.0f} pair slots issued now, {steps_dense:7.0f} if only differing pairs cost")
    prev = st[f"med{i}"]
