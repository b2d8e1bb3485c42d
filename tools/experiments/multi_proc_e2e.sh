#!/bin/bash
# The bench's timed workload (pair set-up from resident raw images + 60 chained frames + writer hand-off per pair) in n PROCESSES side by side,
# each with c contexts and 2 pairs per context per step; aggregate frames/s.  Compare: one process with 3 contexts x 2 pairs = the bench's pool.
cd "$GRAFT_REPO_ROOT"
run() { n=$1; c=$2; steps=$3
  rm -f /tmp/mpe_*.log
  for i in $(seq 1 $n); do python3 tools/experiments/pool_e2e.py $steps $c $((2 * c)) > /tmp/mpe_$i.log 2>&1 & done
  wait
  echo "$n processes x $c contexts: $(cat /tmp/mpe_*.log | grep -o '[0-9.]* frames/s' | awk '{s+=$1; printf "%s ", $1} END {print "-> sum", s}')"
}
run 1 3 30; run 2 1 40; run 3 1 40; run 4 1 40; run 2 2 30; run 3 2 30; run 6 1 30
