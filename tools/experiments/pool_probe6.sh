# kernel traces of the six-context pool with the writer, with the writer's path but no bytes moved (experiments build), and without a writer: which kernels are slower when frames leave?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pool_trace; mkdir -p $O
export POPPY_HIP_LIB=$R/poppy_amd/libpoppy_hip_experiments.so POOL_WITH_TORCH=1
run() { name=$1; shift; env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/$name -o t -- python3 $R/tools/experiments/pool_nowriter.py 2 6 36 ${W:-1} 6 > $O/$name.log 2>&1; tail -1 $O/$name.log; python3 $R/tools/rocprof_summary.py $O/$name/t_results.db | grep -E "k_(warp_bin|unsharp_tile|pyrdown|collapse|tile_expand|pyr_tail|upload|median_cols)|copyBuffer" | awk -F'|' '{printf "    %-28s calls %6s avg %9s total %11s\n", $2, $3, $5, $4}'; }
echo "== writer"; W=1 run writer X=1
echo "== writer, no bytes"; W=1 run nobytes POPPY_DL_SKIP_COPY=1
echo "== no writer"; W=0 run nowriter X=1
