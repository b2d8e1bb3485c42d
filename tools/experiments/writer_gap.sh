#!/bin/bash
# gpurun -- bash tools/experiments/writer_gap.sh     (see writer_gap.py)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for m in resident writer; do
  python3 $R/tools/experiments/writer_gap.py 1920 1080 5 $m
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/wg_$m -o t -- python3 $R/tools/experiments/writer_gap.py 1920 1080 3 $m > /dev/null 2>&1
  python3 $R/tools/experiments/writer_gap.py --db $O/wg_$m/*.db
done
