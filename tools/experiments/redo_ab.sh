# k_gabor_redo, two builds on ONE box (poppy_amd/libpoppy_hip.so and poppy_amd/alt_*.so): the Gabor banks alone (kernel trace averages) and the pair set-up per content
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cp $R/poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2; do for so in /tmp/orig.so $(ls $R/poppy_amd/alt_*.so); do
  cp $so $R/poppy_amd/libpoppy_hip.so
  for kind in photo synthetic textured; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab -o t -- python3 $R/tools/experiments/gabor_alone.py 1920 1080 $kind > /dev/null 2>&1
    echo "$(basename $so) $kind: $(python3 $R/tools/rocprof_summary.py $O/ab/*.db 2>/dev/null | grep -E "k_gabor" | cut -c1-58 | tr '\n' ' ')"; rm -rf $O/ab
  done
  echo "$(basename $so): set-up photo $(python3 $R/tools/experiments/setup_content.py photo 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | synthetic $(python3 $R/tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | textured $(python3 $R/tools/experiments/setup_content.py textured 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done; done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
