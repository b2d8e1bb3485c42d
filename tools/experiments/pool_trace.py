#!/usr/bin/env python3
"""Reads a POPPY_POOL_TRACE file (experiment build): per pool call and worker the times of set-up begin / end (negative) and of every frame written.  Prints, per call:
the frames/s of the OTHER workers while a worker is inside a set-up against their rate outside any set-up, and a coarse timeline."""
import sys, numpy as np
calls = []; cur = []
for line in open(sys.argv[1]):
    if line.startswith("END"): calls.append(cur); cur = []
    elif line.startswith("W"): cur.append([float(x) for x in line.split()[2:]])
calls = calls[5:]                                   # warm-up
rates_in, rates_out, setups = [], [], []
for call in calls:
    t0 = min(abs(x) for w in call for x in w); t1 = max(abs(x) for w in call for x in w)
    wins = []                                       # (worker, begin, end)
    frames = []                                     # (worker, t)
    for k, w in enumerate(call):
        neg = [-x for x in w if x < 0]
        for i in range(0, len(neg) - 1, 2): wins.append((k, neg[i], neg[i + 1]))
        frames += [(k, x) for x in w if x > 0]
    setups += [e - b for _, b, e in wins]
    # time with >= 1 set-up in progress, and frames (of any worker) written in those times
    ev = sorted([(b, 1) for _, b, e in wins] + [(e, -1) for _, b, e in wins])
    t_in = 0.0; depth = 0; last = t0; spans = []
    for t, d in ev:
        if depth > 0: t_in += t - last; spans.append((last, t))
        depth += d; last = t
    ft = np.array(sorted(x for _, x in frames))
    n_in = sum(int(((ft >= a) & (ft < b)).sum()) for a, b in spans)
    rates_in.append((n_in, t_in)); rates_out.append((len(ft) - n_in, (t1 - t0) - t_in))
ni = sum(n for n, _ in rates_in); ti = sum(t for _, t in rates_in); no = sum(n for n, _ in rates_out); to = sum(t for _, t in rates_out)
print(f"{len(calls)} pool calls: a set-up takes {np.mean(setups):.2f} ms (median {np.median(setups):.2f}); time with a set-up in progress {ti / (ti + to) * 100:.0f} % of the call")
print(f"frames written while some worker sets a pair up: {ni / ti * 1e3:.0f} frames/s; while none does: {no / to * 1e3:.0f} frames/s; overall {(ni + no) / (ti + to) * 1e3:.0f}")
