#!/bin/bash
# gpurun -- bash tools/experiments/hwq_sweep.sh : frames/s with writer against GPU_MAX_HW_QUEUES, chained and phase mode, 1080p
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for q in 1 2 3 4 5 6 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  f() { python3 $R/tools/experiments/writer_gap.py 1920 1080 3 $1 $2 | cut -d: -f2 | cut -d, -f1; }
  echo "queues $q: plain chain$(f writer) phase$(f writer-phase) | torch chain$(f writer torch) phase$(f writer-phase torch) | resident chain$(f resident) phase$(f resident-phase)"
done
