#!/bin/bash
# k_unsharp_stream with different segment heights (POPPY_UNSHARP_ROWS), rocprofv3 kernel-trace average in the chained frame loop.
# Usage: gpurun -- bash tools/experiments/unsharp_rows.sh "8 12 16 24 32"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for rows in ${1:-0 8 12 16 24 32 48}; do
  for sz in "1920 1080" "3840 2160"; do set -- $sz
    POPPY_UNSHARP_ROWS=$rows timeout 300 rocprofv3 --kernel-trace --stats -d $O/ur -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 2 > $O/ur.log 2>&1
    echo "rows $rows $1: $(python3 $R/tools/rocprof_summary.py $O/ur/*.db 2>/dev/null | grep -E "unsharp_stream" | head -1) $(grep frames/s $O/ur.log)"; rm -rf $O/ur
  done
done
