#!/bin/bash
# gpurun -- bash tools/experiments/setup_trace_4k.sh : kernel times of the 3840x2160 pair set-up (6 set-ups after 2 warm ones)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
python3 $R/tools/experiments/setup_q.py
timeout 600 rocprofv3 --kernel-trace --stats -d $O/su4k -o t -- python3 $R/tools/experiments/setup_q.py > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$O/su4k/*.db")[0])
rows = list(db.execute("select name, start, end, grid_x from kernels order by start"))
# the 4K set-ups: everything after the first 4K-sized k_bgr2gray / after half of the trace
big = [r for r in rows if r[3] >= 3840 * 2160 // 4 or True]
t_half = rows[len(rows) // 2][1]
agg = {}
for n, s, e, g in rows:
    if s < t_half: continue
    k = n.split('(')[0].replace('poppy_hip::', '').replace('void ', '')
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"second half of the trace (mostly the 4K set-ups): {tot / 1e6:.1f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:28]:
    print(f"  {k:40s} {v[0]:5d} calls {v[1] / 1e3:10.1f} us total {v[1] / v[0] / 1e3:9.1f} us avg")
PY
