cd $GRAFT_REPO_ROOT
for rep in 1 2; do for rows in 32 64 48 24; do
  echo "rows=$rows: synthetic $(POPPY_MED_COLS_ROWS=$rows python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | textured(forced cols) $(POPPY_MED_COLS_FORCE=1 POPPY_MED_COLS_ROWS=$rows python3 tools/experiments/setup_content.py textured 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
