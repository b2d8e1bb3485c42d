cd $GRAFT_REPO_ROOT
for rs in "2 0" "0 1" "2 1" "3 1" "2 2" "0 2"; do python3 tools/experiments/setup_interference.py $rs 3 2>&1 | tail -1; done
