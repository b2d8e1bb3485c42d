# EXPERIMENT (timing build, results wrong): is it the medians' OCCUPANCY of the GPU or their LENGTH that a set-up costs the pool?
# POPPY_X_MED=skip: median launches cost nothing; =sleep: each takes its usual time on one wave, the GPU otherwise free
cd $GRAFT_REPO_ROOT
for p in "X=0" "POPPY_X_MED=sleep" "POPPY_X_MED=skip" "X=0" "POPPY_X_MED=sleep"; do
  echo "[$p] pool e2e $(env $p timeout 300 python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | $(env $p timeout 300 python3 tools/experiments/setup_interference.py 2 1 3 2>&1 | tail -1) | alone $(env $p timeout 300 python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done
