cd $GRAFT_REPO_ROOT
for p in "" "POPPY_SETUP_PRIO=low" "" "POPPY_SETUP_PRIO=low"; do
  echo "[$p] $(env $p python3 tools/experiments/setup_interference.py 2 1 3 2>&1 | tail -1) | $(env $p python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | alone $(env $p python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done
