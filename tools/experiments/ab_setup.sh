# A/B of two builds of the library (poppy_amd/altA.so, altB.so) on one box: the pair set-up alone, three side by side, the pool; median kernel times
cd $GRAFT_REPO_ROOT
cp poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2; do for v in A B; do
  cp poppy_amd/alt$v.so poppy_amd/libpoppy_hip.so
  echo "build $v: alone $(python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | photo $(python3 tools/experiments/setup_content.py photo 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | $(python3 tools/experiments/setup_interference.py 0 3 3 2>&1 | tail -1 | grep -o '[0-9.]* set-ups/s') | pool $(python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s')"
done; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for v in A B; do
  cp $R/poppy_amd/alt$v.so $R/poppy_amd/libpoppy_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab -o t -- python3 $R/tools/experiments/setup_content.py synthetic 1920 1080 5 > /dev/null 2>&1
  echo "build $v medians: $(python3 $R/tools/rocprof_summary.py $O/ab/*.db 2>/dev/null | grep k_median | awk -F'|' '{s+=$5} END {print s}') us per chain"; rm -rf $O/ab
done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
