#!/usr/bin/env python3
"""Kernel trace (rocprofv3 results.db) of contexts that only set pairs up: how many kernels run at a time, how busy each hardware queue is, what runs beside what.
usage: phase_concurrency.py results.db"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns of `kernels`:", cols)
qcol = next((c for c in cols if "queue" in c.lower()), None)
scol = next((c for c in cols if "stream" in c.lower()), None)
sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
rows = list(db.execute(f"select {sel} from kernels order by start"))
t0, t1 = rows[len(rows) // 4][1], rows[3 * len(rows) // 4][1]          # the middle half of the run
rows = [r for r in rows if r[1] >= t0 and r[2] <= t1]
span = (t1 - t0) / 1e3
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
hist = collections.Counter(); depth = 0; last = t0
for t, d in ev:
    hist[depth] += t - last; depth += d; last = t
tot = sum(hist.values())
print(f"window {span / 1e3:.1f} ms, {len(rows)} kernels; kernels in flight -> share of time: " + ", ".join(f"{k}: {v / tot * 100:.1f} %" for k, v in sorted(hist.items())))
busy = collections.defaultdict(float); names = collections.defaultdict(collections.Counter)
for n, a, b, q, s in rows:
    busy[(q, s)] += b - a
    names[(q, s)][n.split("(")[0].replace("poppy_hip::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:24]] += 1
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"queue {k[0]} stream {k[1]}: busy {v / (t1 - t0) * 100:5.1f} %  ({', '.join(f'{a} x{c}' for a, c in names[k].most_common(3))})")
ksum = collections.defaultdict(float)
for n, a, b, q, s in rows: ksum["median" if "median" in n else "gabor" if "gabor" in n else "other"] += b - a
print("kernel time per class / window: " + ", ".join(f"{k} {v / (t1 - t0):.2f}" for k, v in ksum.items()))
