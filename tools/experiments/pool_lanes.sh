# EXPERIMENT: a pool whose contexts take their streams from four lanes (one hardware queue each) instead of making their own
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for p in "X=0" "POPPY_POOL_LANES=1" "POPPY_POOL_LANES=1 POPPY_POOL_AUX=side" "POPPY_POOL_LANES=1 POPPY_POOL_AUX=next"; do
  echo "[$p] pool e2e (no torch) $(env $p timeout 300 python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | bench (torch) $(env $p timeout 300 python3 bench.py --headline-only --steps 30 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
done; done
