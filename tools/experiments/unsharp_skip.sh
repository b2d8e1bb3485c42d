# what the phases of k_unsharp_tile cost: the experiments build with phases left out (POPPY_UNSHARP_SKIP bits: 1 row pass, 2 column pass, 4 the last phase's arithmetic,
# 8 the source loads, 16 the stores; 63 the launch alone; wrong frames)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export POPPY_HIP_LIB=$R/poppy_amd/libpoppy_hip_experiments.so
for sk in 0 1 2 4 8 16 7 24 31 63 0; do
export POPPY_UNSHARP_SKIP=$sk
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/us_$sk -o t -- python3 $R/tools/experiments/frames_only.py 1920 1080 60 chain 2 > /dev/null 2>&1
echo "skip $sk: $(python3 $R/tools/rocprof_summary.py $R/gpurun_out/us_$sk/t_results.db | grep -E "k_unsharp_tile" | awk -F'|' '{print $5}')"; rm -rf $R/gpurun_out/us_$sk
done
