# round 4: what the exact form of the FFT Gabor banks (doubtful pixels re-formed as direct sums) costs the pair set-up: A/B in one process environment each, no profiler
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for k in synthetic photo textured; do
  a=$(python3 tools/experiments/setup_content.py $k 1920 1080 25 2>&1 | tail -1)
  b=$(POPPY_HIP_LIB=$PWD/poppy_amd/libpoppy_hip_experiments.so POPPY_GABOR_NO_REDO=1 python3 tools/experiments/setup_content.py $k 1920 1080 25 2>&1 | tail -1)
  echo "with redo:    $a"; echo "without redo: $b"
done; done
