#!/usr/bin/env python3
"""Where a wave of the fused warp kernel spends its life (poppy_hip_warp_probe: k_warp_bin with fenced, s_memtime-stamped phases).
Prints per size: the phases' durations in shader cycles (median / mean / p90), the kernel's span in cycles against its wall time (the clock the
part sustains), how many waves per SIMD are in an issuing phase (map arithmetic, blend) on average, and the slots' idle time between workgroups.
usage: warp_probe.py [W H]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from poppy_amd import capi, synth
sizes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(1920, 1080), (3840, 2160)]
for w, h in sizes:
    a, b = synth.gen_pair(w, h, seed=1234)
    c = capi.Context(0, number_of_frames=60)
    c.pair_begin(a, b)
    for t in (0.13, 0.02):
        c.render(t, t)
        if c.last_warp_kind() == 2:
            break
    base_us = sorted(c.time_last_warp(50) * 1e3 for _ in range(3))[1]
    st, ms = c.warp_probe()
    st = st.astype(np.int64)
    T = st[:, :6]
    hw, xcc = st[:, 6], st[:, 7] & 15
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    names = ["front end (ids + records + barrier)", "map arithmetic (4 px x 2 sources)", "gather issue", "gather wait", "blend + store issue"]
    d = np.diff(T, axis=1)
    print(f"== {w}x{h}: k_warp_bin relaunched {base_us:.1f} us; probe launch {ms * 1e3:.1f} us, {len(T)} waves")
    for i, n in enumerate(names):
        print(f"  {n:42s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  p90 {np.percentile(d[:, i], 90):8.0f} cycles")
    life = T[:, 5] - T[:, 0]
    print(f"  {'wave lifetime':42s} median {np.median(life):8.0f}  mean {life.mean():8.0f}  p90 {np.percentile(life, 90):8.0f} cycles")
    # clock: per XCC the span first entry -> last end, in counter ticks, against the launch's wall time
    for x in sorted(set(xcc.tolist()))[:2]:
        m = xcc == x
        span = T[m, 5].max() - T[m, 0].min()
        print(f"  XCC {x}: {m.sum()} waves, span {span} ticks = {span / (ms * 1e3):.0f} ticks/us of the launch")
    # per SIMD: sum of the issuing phases against the span it was busy for
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    key = key * 4 + simd
    uk, inv = np.unique(key, return_inverse=True)
    issue = d[:, 1] + d[:, 4]
    per = np.zeros(len(uk)); first = np.full(len(uk), np.iinfo(np.int64).max); last = np.zeros(len(uk), np.int64); cnt = np.zeros(len(uk))
    np.add.at(per, inv, issue); np.minimum.at(first, inv, T[:, 0]); np.maximum.at(last, inv, T[:, 5]); np.add.at(cnt, inv, 1)
    np.add.at(cnt, inv, 0)
    occ_all = np.zeros(len(uk)); np.add.at(occ_all, inv, life)
    span = (last - first).astype(np.float64)
    print(f"  {len(uk)} SIMDs seen, {cnt.mean():.1f} waves each; per SIMD: span {span.mean():.0f} ticks, "
          f"waves resident on average {np.mean(occ_all / span):.2f}, waves in an issuing phase on average {np.mean(per / span):.2f}")
    del c
