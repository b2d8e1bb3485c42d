# pair set-up, two builds on ONE box (libpoppy_hip.so and poppy_amd/alt_*.so), many alternations: synthetic 1080p only, median of 25 set-ups each
R="$GRAFT_REPO_ROOT"; cd $R
cp poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2 3 4 5; do
for so in /tmp/orig.so $(ls poppy_amd/alt_*.so); do
  cp $so poppy_amd/libpoppy_hip.so
  echo "$(basename $so): synthetic $(python3 tools/experiments/setup_content.py synthetic 1920 1080 25 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms (min [0-9.]*)') | photo $(python3 tools/experiments/setup_content.py photo 1920 1080 25 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms (min [0-9.]*)')"
done; done
cp /tmp/orig.so poppy_amd/libpoppy_hip.so
