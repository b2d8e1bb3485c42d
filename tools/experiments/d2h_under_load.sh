#!/bin/bash
# Does the device-to-host copy path slow down when kernels run beside it?  A copy-only process beside 0 / 1 / 3 processes rendering resident chained frames.
cd "$GRAFT_REPO_ROOT"
python3 tools/experiments/d2h_under_load.py 2
for n in 1 3; do
  for i in $(seq 1 $n); do python3 tools/experiments/frames_only.py 1920 1080 60 chain 1200 > /tmp/dl_$i.log 2>&1 & done
  sleep 5
  echo "beside $n rendering process(es): $(python3 tools/experiments/d2h_under_load.py 2.0)"
  wait
  echo "   renderers: $(cat /tmp/dl_*.log | grep -o '[0-9.]* frames/s' | tr '\n' ' ')"
done
