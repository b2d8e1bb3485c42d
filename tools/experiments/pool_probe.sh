cd $GRAFT_REPO_ROOT
for c in 2 3 4 6; do for w in 0 1; do timeout 300 python3 tools/experiments/pool_nowriter.py 6 $c 6 $w | tail -1; done; done
POPPY_HIP_NOCONE=1 timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1
POPPY_HIP_NOCONE=1 timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 0 | tail -1
