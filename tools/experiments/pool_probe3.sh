cd $GRAFT_REPO_ROOT
for w in 0 1; do echo "== writer $w"; POPPY_SETUP_TIMING=1 POPPY_SEQ_TIMING=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 6 $w 2>&1 | grep -E "pair set-up|sequence of|frames/s" | tail -13; done
