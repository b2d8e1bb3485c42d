# A/B of the Gabor FFT kernels alone on one box: poppy_amd/altA.so, altB.so (see ab_lib.sh)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cp $R/poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2; do for v in A B; do
  cp $R/poppy_amd/alt$v.so $R/poppy_amd/libpoppy_hip.so
  for kind in photo synthetic; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab -o t -- python3 $R/tools/experiments/gabor_alone.py 1920 1080 $kind > /dev/null 2>&1
    echo "build $v $kind: $(python3 $R/tools/rocprof_summary.py $O/ab/*.db 2>/dev/null | grep -E "k_gabor" | cut -c1-58 | tr '\n' ' ')"; rm -rf $O/ab
  done
done; done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
