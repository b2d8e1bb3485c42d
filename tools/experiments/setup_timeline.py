#!/usr/bin/env python3
"""Kernel timeline of the LAST pair set-up in a rocprofv3 kernel trace of tools/experiments/pair_begin_time.py (start, end, duration in us,
stream, kernel), from the first kernel after the previous set-up's k_gray_inv to this one's.   python tools/experiments/setup_timeline.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,duration,stream_id from kernels order by start"))
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "")
ends = [i for i, r in enumerate(rows) if 'k_gray_inv' in r[0]]
i0, i1 = ends[-2] + 1, ends[-1]
while 'k_upload' in rows[i0][0] or 'k_tile_expand' in rows[i0][0] or 'k_warp' in rows[i0][0] or 'pyr' in rows[i0][0] or 'collapse' in rows[i0][0] or 'unsharp_t' in rows[i0][0]: i0 += 1
t0 = rows[i0][1]
busy = 0
for r in rows[i0:i1 + 1]:
    print(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-t0)/1e3:8.1f} {r[3]/1e3:7.1f}  s{r[4]}  {short(r[0])}")
