# from which window on images of many values per tile (photographs, texture) take the column-histogram median (POPPY_MED_COLS_MIN_HARD; 999 = never)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for min in 999 33 41 49 57 65 73; do
  echo "hard min=$min: photo $(POPPY_MED_COLS_MIN_HARD=$min python3 tools/experiments/setup_content.py photo 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | textured $(POPPY_MED_COLS_MIN_HARD=$min python3 tools/experiments/setup_content.py textured 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | photo 4K $(POPPY_MED_COLS_MIN_HARD=$min python3 tools/experiments/setup_content.py photo 3840 2160 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
