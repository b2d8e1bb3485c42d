cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for rep in 1 2 3; do for v in 0 1; do
  export POPPY_TILE_EXPAND_PAIR=$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab -o t -- python3 $R/tools/experiments/frames_only.py 3840 2160 60 phase 3 > /dev/null 2>&1
  echo "pair=$v 4K phase: $(python3 $R/tools/rocprof_summary.py $O/ab/*.db 2>/dev/null | grep -E "k_tile_expand" | head -1)"; rm -rf $O/ab
done; done
