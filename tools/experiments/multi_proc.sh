#!/bin/bash
# Do independent chained sequences of SEVERAL PROCESSES share the GPU better than several contexts of one process (the pool: 3 contexts = 1 context's rate)?
# n copies of frames_only.py (resident pair, 60 chained 1080p frames, frames left in HBM) side by side; aggregate frames/s.
cd "$GRAFT_REPO_ROOT"
for n in 1 2 3 4; do
  rm -f /tmp/mp_*.log
  for i in $(seq 1 $n); do python3 tools/experiments/frames_only.py 1920 1080 60 chain 40 > /tmp/mp_$i.log 2>&1 & done
  wait
  echo "$n processes: $(cat /tmp/mp_*.log | grep -o '[0-9.]* frames/s' | awk '{s+=$1; printf "%s ", $1} END {print "-> sum", s}')"
done
