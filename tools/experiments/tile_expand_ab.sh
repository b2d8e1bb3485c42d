# k_tile_expand / k_warp_bin kernel times in the frame loop, chained and phase mode, 1080p and 4K (kernel trace)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/te_ab"
rm -rf $O; mkdir -p $O
for cfg in "1920 1080 chain" "3840 2160 chain" "3840 2160 phase"; do set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 $3 3 > $O/log.txt 2>&1
  echo "== $cfg: $(tail -1 $O/log.txt)"; python3 $R/tools/rocprof_summary.py $O/t/*.db | grep -E "tile_expand|warp_bin"; rm -rf $O/t
done
