# where gabor2 starts (POPPY_GABOR2_AT: 0 behind the second image's medians, 1 / 2 behind the first image's ORB input / FAST kernels): pair set-up per content
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for at in 0 1 2; do
  echo "gabor2_at=$at: synthetic $(POPPY_GABOR2_AT=$at python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | photo $(POPPY_GABOR2_AT=$at python3 tools/experiments/setup_content.py photo 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | 4K $(POPPY_GABOR2_AT=$at python3 tools/experiments/setup_content.py synthetic 3840 2160 9 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
