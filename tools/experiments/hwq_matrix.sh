#!/bin/bash
# gpurun -- bash tools/experiments/hwq_matrix.sh : GPU_MAX_HW_QUEUES 4 (the runtime's default) against 8, every frame-loop mode, without and with torch in the process
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  for t in "" torch; do for sz in "1920 1080" "3840 2160"; do for m in resident writer resident-phase writer-phase; do
    echo "queues $q, ${t:-plain}: $(python3 $R/tools/experiments/writer_gap.py $sz 3 $m $t)"
  done; done; done
  echo "queues $q, pool without torch: $(python3 $R/tools/experiments/pool_e2e.py 2>&1 | tail -1)"
  cd $R; echo "queues $q, bench: $(python3 bench.py --headline-only 2>/dev/null | tail -1 | cut -c1-130)"; cd /tmp
done
