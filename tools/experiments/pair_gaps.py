#!/usr/bin/env python3
"""Where a pair's time goes in a kernel trace of tools/experiments/pool_e2e.py with ONE context (pairs one after the other): per pair the span of
its set-up kernels, the span of its frame kernels and the gaps between them (us).   python tools/experiments/pair_gaps.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
def kind(n):
    for k in ("k_warp", "k_pyr", "k_collapse", "k_unsharp_tile", "k_unsharp_stream", "k_tile_expand", "k_upload", "k_dissolve"):
        if k in n: return "frame"
    if "rocclr" in n: return "copy"
    return "setup"
spans = []          # (kind, start, end, launches)
for n, s, e in rows:
    k = kind(n)
    if k == "copy": continue
    if spans and spans[-1][0] == k: spans[-1][2] = max(spans[-1][2], e); spans[-1][3] += 1
    else: spans.append([k, s, e, 1])
t0 = spans[0][1]
prev_end = None
for k, s, e, n in spans[-14:]:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{k:6s} start {(s-t0)/1e3:10.1f}  span {(e-s)/1e3:9.1f} us  {n:5d} launches   gap before {gap:8.1f} us")
    prev_end = e
