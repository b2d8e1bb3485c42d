#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (results.db) of frames rendered beside pair set-ups: for every frame kernel, its average duration when it runs beside a set-up kernel
(by class: median / gabor / other set-up) and when it does not; and how much of the set-up kernels' time has a frame kernel beside it.   usage: overlap_stats.py results.db"""
import sqlite3, sys, bisect
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
FRAME = ("k_warp_bin", "k_pyrdown", "k_pyr_tail", "k_collapse", "k_unsharp_tile", "k_unsharp_stream", "k_tile_expand", "k_upload")
def cls(n):
    s = n.replace("poppy_hip::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if s.startswith(FRAME): return "frame", s.split("<")[0]
    if "median" in s: return "median", s.split("<")[0]
    if "gabor" in s: return "gabor", s.split("<")[0]
    if "copyBuffer" in s or "rocclr" in s: return "copy", s
    return "setup", s.split("<")[0]
items = [(cls(n), a, b) for n, a, b in rows]
setup_iv = {"median": [], "gabor": [], "setup": []}
for (c, s), a, b in items:
    if c in setup_iv: setup_iv[c].append((a, b))
def overlap(ivs, a, b):
    t = 0
    for x, y in ivs:
        if y <= a: continue
        if x >= b: break
        t += min(b, y) - max(a, x)
    return t
for k in setup_iv: setup_iv[k].sort()
stats = {}
for (c, s), a, b in items:
    if c != "frame": continue
    d = b - a
    ov = {k: overlap(v, a, b) for k, v in setup_iv.items()}
    kind = max(ov, key=ov.get) if max(ov.values()) > 0.5 * d else "alone"
    e = stats.setdefault(s, {}).setdefault(kind, [0, 0.0]); e[0] += 1; e[1] += d
print("frame kernel: average duration (us) [launches] by what ran beside it for more than half of its time")
for s, d in sorted(stats.items()):
    print(f"  {s:24s} " + "  ".join(f"{k}: {v[1] / v[0] / 1e3:7.1f} [{v[0]}]" for k, v in sorted(d.items())))
frame_iv = sorted((a, b) for (c, s), a, b in items if c == "frame")
for k, v in setup_iv.items():
    tot = sum(b - a for a, b in v); cov = sum(overlap(frame_iv, a, b) for a, b in v)
    if tot: print(f"{k:7s} kernels: {tot / 1e6:.1f} ms in total, frame kernels beside them for {100.0 * cov / tot:.0f} % of that (summed over frame kernels: can exceed 100)")
