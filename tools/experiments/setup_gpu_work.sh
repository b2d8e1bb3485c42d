#!/bin/bash
# GPU work of one pair set-up with nothing beside it: kernel trace of set-ups whose two chains run one after the other (POPPY_SETUP_SERIAL=1: kernel times close to stand-alone),
# summed per kernel and divided by the number of set-ups.  Usage: gpurun -- bash tools/experiments/setup_gpu_work.sh {synthetic|photo|textured} [W H]
kind=${1:-synthetic}; W=${2:-1920}; H=${3:-1080}
cd /tmp && export TMPDIR=/tmp
export POPPY_SETUP_SERIAL=1
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
rm -rf $O/gpu_work
timeout 600 rocprofv3 --kernel-trace --stats -d $O/gpu_work -o t -- python3 $R/tools/experiments/setup_content.py $kind $W $H 9 > $O/gpu_work_log.txt 2>&1
python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$O/gpu_work/*.db")[0])
rows = list(db.execute("select name, count(*), sum(duration) from kernels group by name order by sum(duration) desc"))
n = 10.0                                   # 1 + 9 set-ups in the run
tot = sum(r[2] for r in rows) / n / 1e3
print(f"GPU work per set-up ($kind ${W}x$H, chains one after the other): {tot:.0f} us of kernels")
for name, cnt, dur in rows[:28]:
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "")
    print(f"  {dur / n / 1e3:8.1f} us  {cnt / n:5.1f} x {dur / cnt / 1e3:7.1f}  {short}")
PY
tail -1 $O/gpu_work_log.txt
