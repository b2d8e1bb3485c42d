#!/bin/bash
# Kernel timeline of one pair set-up on the named content (rocprofv3 kernel trace of tools/experiments/setup_content.py; tools/experiments/setup_timeline.py prints it).
# Usage: gpurun -- bash tools/experiments/setup_timeline.sh <tag> {synthetic|photo|textured} [W H]
tag=${1:-tl}; kind=${2:-photo}; W=${3:-1920}; H=${4:-1080}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_trace -o t -- python3 $R/tools/experiments/setup_content.py $kind $W $H 4 > $O/${tag}_log.txt 2>&1
python3 $R/tools/experiments/setup_timeline.py $O/${tag}_trace/*.db > $O/${tag}_timeline.txt 2>&1
python3 $R/tools/rocprof_summary.py $O/${tag}_trace/*.db 2>/dev/null | head -30
