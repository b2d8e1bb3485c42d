# rows per tile of the column-histogram median (POPPY_MED_COLS_ROWS; default: about 512 tiles per image), pair set-up at 1080p and 4K
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for rows in 0 24 32 48 64 96; do
  echo "rows=$rows: synthetic $(POPPY_MED_COLS_ROWS=$rows python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | 4K $(POPPY_MED_COLS_ROWS=$rows python3 tools/experiments/setup_content.py synthetic 3840 2160 9 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
