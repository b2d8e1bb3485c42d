#!/bin/bash
# Per-kernel averages of the chained frame loop at 1080p and 4K (rocprofv3 --kernel-trace --stats) plus the untraced frame rates.
# Usage: gpurun -- bash tools/experiments/kernel_times.sh <tag> [kernel-name-filter]
tag=${1:-kt}; filt=${2:-k_}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
F="python3 $R/tools/experiments/frames_only.py"
for sz in "1920 1080 5" "3840 2160 3"; do set -- $sz
  $F $1 $2 60 chain 10; $F $1 $2 60 phase 10
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_$1 -o t -- $F $1 $2 60 chain $3 > /dev/null 2>&1
  python3 $R/tools/rocprof_summary.py $O/${tag}_$1/*.db 2>/dev/null | grep -E "$filt" | head -14
done
