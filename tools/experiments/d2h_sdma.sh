#!/bin/bash
# Does the frame hand-off (device -> pinned host) run on the SDMA engines or as a blit KERNEL on the CUs?
# Kernel trace + memory-copy trace of (a) the raw copy probe with plain and with "mapped" pinned memory, (b) the library's writer path.
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
show() { python3 - "$1" <<PY
import sqlite3, glob, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
rows = list(db.execute("select name, total_calls, average from top_kernels where name like '%copyBuffer%' or name like '%rocclr%'"))
print("  blit kernels:", rows if rows else "none")
try: print("  memory_copies rows:", list(db.execute("select count(*) from memory_copies"))[0][0])
except Exception as e: print("  ", e)
PY
}
for flags in 0 2; do
  echo "== raw probe, hipHostMalloc flags $flags"
  D2H_HOST_FLAGS=$flags timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/d2h_raw$flags -o t -- python3 $R/tools/experiments/d2h_raw.py 2>/dev/null | head -1
  show $O/d2h_raw$flags; rm -rf $O/d2h_raw$flags
done
echo "== library: pair_begin + 60 chained frames with a writer"
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/d2h_lib -o t -- python3 $R/tools/experiments/pair_begin_time.py 2>/dev/null | tail -1
show $O/d2h_lib; rm -rf $O/d2h_lib
