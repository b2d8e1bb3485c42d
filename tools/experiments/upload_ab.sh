# pair set-up with the second image uploaded beside the first image's chain (default) or both uploads first (POPPY_SETUP_UPLOAD_BOTH=1)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for both in 0 1; do
  if [ $both = 1 ]; then export POPPY_SETUP_UPLOAD_BOTH=1; else unset POPPY_SETUP_UPLOAD_BOTH; fi
  echo "upload_both=$both: synthetic $(python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | photo $(python3 tools/experiments/setup_content.py photo 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | 4K $(python3 tools/experiments/setup_content.py synthetic 3840 2160 9 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
