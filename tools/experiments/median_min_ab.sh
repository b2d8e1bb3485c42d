# from which window on images of few values per tile take the column histograms (POPPY_MED_COLS_MIN; default 25): pair set-up, median of 25
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for min in 25 17 9; do
  echo "min=$min: synthetic $(POPPY_MED_COLS_MIN=$min python3 tools/experiments/setup_content.py synthetic 1920 1080 25 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | 4K $(POPPY_MED_COLS_MIN=$min python3 tools/experiments/setup_content.py synthetic 3840 2160 9 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
