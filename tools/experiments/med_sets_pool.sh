#!/bin/bash
# Histogram sets per median launch against POOL throughput: fewer sets = longer segments = less warm-up work in total and less of the GPU's LDS per launch,
# at the price of a longer single set-up.  pool_e2e.py (3 contexts x 2 pairs, set-up + 60 chained frames + writer) and pair_begin latency.
cd "$GRAFT_REPO_ROOT"
for sets in 1024 768 512 384 256 1024 512; do
  echo "sets $sets: $(POPPY_MED_SETS=$sets python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s.*') | set-up alone $(POPPY_MED_SETS=$sets python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done
