cd $GRAFT_REPO_ROOT
for p in "X=0" "POPPY_CTX_LANES=1" "X=0" "POPPY_CTX_LANES=1"; do
  echo "[$p] $(env $p timeout 300 python3 tools/experiments/setup_interference.py 2 1 3 2>&1 | tail -1) | $(env $p timeout 300 python3 tools/experiments/setup_interference.py 2 0 3 2>&1 | tail -1) | alone $(env $p timeout 300 python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done
