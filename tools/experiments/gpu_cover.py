#!/usr/bin/env python3
"""How busy the GPU is in a kernel trace: the part of the traced span during which at least one kernel runs, the average number of kernels
running at once, and the kernels by their share of (a) summed duration, (b) time during which they run ALONE.
   python tools/experiments/gpu_cover.py <results.db> [skip_first_fraction]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = list(db.execute("select name,start,end from kernels order by start"))
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "").split("<")[0]
t_lo = rows[0][1] + (rows[-1][2] - rows[0][1]) * skip
rows = [r for r in rows if r[1] >= t_lo]
ev = []
for i, (n, s, e) in enumerate(rows): ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]; busy = 0; area = 0; alone = defaultdict(float)
for t, d, i in ev:
    dt = t - last
    if active:
        busy += dt; area += dt * len(active)
        if len(active) == 1: alone[short(rows[next(iter(active))][0])] += dt
    last = t
    if d == 1: active.add(i)
    else: active.discard(i)
span = rows[-1][2] - rows[0][1]
tot = defaultdict(float)
for n, s, e in rows: tot[short(n)] += e - s
print(f"span {span/1e6:.1f} ms, some kernel running {100*busy/span:.1f} %, average kernels at once while busy {area/busy:.2f}, idle {100*(1-busy/span):.1f} %")
print("kernel: share of summed kernel time | share of the span it runs alone")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:22]:
    print(f"  {k:34s} {100*v/sum(tot.values()):5.1f} %   {100*alone[k]/span:5.1f} %")
