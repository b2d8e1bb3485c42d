#!/bin/bash
# gpurun -- bash tools/experiments/hwq_sweep4.sh : as hwq_sweep.sh, plus the pooled end-to-end rates with 4 and 8 queues
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
# (the shipped library: download stream created third; the "lazy" rows of r02_notes section 7 came from the build before that)
for q in 4 5 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  f() { python3 $R/tools/experiments/writer_gap.py 1920 1080 3 $1 $2 | cut -d: -f2 | cut -d, -f1; }
  echo "download stream created third, queues $q: plain chain$(f writer) phase$(f writer-phase) | torch chain$(f writer torch) phase$(f writer-phase torch)"
done
for q in 4 8; do export GPU_MAX_HW_QUEUES=$q
  echo "queues $q, pool without torch: $(python3 $R/tools/experiments/pool_e2e.py 2>&1 | tail -1)"
  cd $R; echo "queues $q, bench: $(python3 bench.py --headline-only 2>/dev/null | tail -1 | cut -c1-130)"; cd /tmp
done
