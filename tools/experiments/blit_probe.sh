#!/bin/bash
# Which path do the frame downloads of the pooled bench take (SDMA copy or blit kernel on the CUs)?  bench.py --headline-only under
# the kernel + memory-copy trace, for a few settings.
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
show() { python3 - "$1" <<PY
import sqlite3, glob, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
big = list(db.execute("select count(*), avg(end-start)/1000.0 from kernels where name like '%rocclr_copyBuffer%' and end-start > 30000"))
print("  blit kernels > 30 us:", big, " memory_copies rows:", list(db.execute("select count(*) from memory_copies"))[0][0])
PY
}
run() { echo "== $1"; shift; env "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/bp -o t -- python3 $R/bench.py --steps 3 --warmup 1 --headline-only $EXTRA 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  value', d['value'])"; show $O/bp; rm -rf $O/bp; }
EXTRA="" run "default (2 contexts, 8 hw queues)" X=1
EXTRA="--contexts 1" run "1 context" X=1
EXTRA="" run "GPU_MAX_HW_QUEUES=4" GPU_MAX_HW_QUEUES=4
EXTRA="" run "HSA_ENABLE_SDMA=1" HSA_ENABLE_SDMA=1
