#!/bin/bash
# Where the pair set-up's time goes: host stage stamps (POPPY_SETUP_TIMING) and the kernel trace of four set-ups.
# Usage: gpurun -- bash tools/experiments/setup_stages.sh <tag>
tag=${1:-setup}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
POPPY_SETUP_TIMING=1 python3 $R/tools/experiments/pair_begin_time.py 2>&1 | tail -8
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_trace -o t -- python3 $R/tools/experiments/pair_begin_time.py > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $O/${tag}_trace/*.db 2>/dev/null | grep -v "k_warp\|k_pyr\|k_collapse\|k_unsharp_t\|k_tile\|k_upload" | head -60
