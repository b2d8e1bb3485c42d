#!/bin/bash
# Kernel times of the pair set-up (rocprofv3 --kernel-trace --stats of tools/experiments/pair_begin_time.py) plus the untraced set-up time.
# Usage: gpurun -- bash tools/experiments/setup_kernels.sh <tag> [kernel-name-filter]
tag=${1:-su}; filt=${2:-k_}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
python3 $R/tools/experiments/pair_begin_time.py 2>&1 | grep pair_begin
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_setup -o t -- python3 $R/tools/experiments/pair_begin_time.py > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $O/${tag}_setup/*.db 2>/dev/null | grep -E "$filt" | head -24
