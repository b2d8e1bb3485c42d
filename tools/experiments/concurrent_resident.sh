cd $GRAFT_REPO_ROOT
for k in 1 2 3 4; do python3 tools/experiments/concurrent_resident.py $k 20 2>&1 | tail -2; done
