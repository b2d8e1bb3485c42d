cd $GRAFT_REPO_ROOT
for w in 0 4 2 0 4; do
  e=""; [ $w != 0 ] && e="POPPY_MED_WAVES=$w"
  echo "[waves ${w}] pool $(env $e python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | 2 render + 1 set-up: $(env $e python3 tools/experiments/setup_interference.py 2 1 2 2>&1 | tail -1 | grep -o '[0-9]* frames/s, [0-9.]* set-ups/s') | alone $(env $e python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done
