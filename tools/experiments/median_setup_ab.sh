# pair set-up with the medians by one kernel or the other (POPPY_MED_COLS_FORCE=0: a lane per column everywhere; default: by content), three contents
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for force in 0 -1 1; do
  if [ $force = -1 ]; then unset POPPY_MED_COLS_FORCE; else export POPPY_MED_COLS_FORCE=$force; fi
  echo "force=$force min=${POPPY_MED_COLS_MIN:-25}: synthetic $(python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | photo $(python3 tools/experiments/setup_content.py photo 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | textured $(python3 tools/experiments/setup_content.py textured 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
unset POPPY_MED_COLS_FORCE
for min in 9 17 33 41; do echo "min=$min: synthetic $(POPPY_MED_COLS_MIN=$min python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"; done
