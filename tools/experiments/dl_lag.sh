#!/bin/bash
# POPPY_HIP_DL_LAG: the host issues a frame's download one (two) frames later, so that it never blocks on the newest frame.  Chained 1080p with a writer:
# one context on a resident pair (writer_rate.py) and the bench's pool (pool_e2e.py).
cd "$GRAFT_REPO_ROOT"
for lag in 0 1 2 0 1; do
  echo "lag $lag: $(POPPY_HIP_DL_LAG=$lag python3 tools/experiments/writer_rate.py 2>&1 | tail -1) | $(POPPY_HIP_DL_LAG=$lag python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1)"
done
