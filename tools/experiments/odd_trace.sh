# which kernels an odd width costs: kernel traces of the chained loop at 749 / 752 x 480 and 3838 / 3840 x 2160
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for sz in "749 480" "752 480" "3838 2160" "3840 2160"; do w=${sz% *}
echo "== $sz: $(python3 $R/tools/experiments/frames_only.py $sz 60 chain 5 | tail -1)"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/odd_trace_$w -o t -- python3 $R/tools/experiments/frames_only.py $sz 60 chain 3 > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $R/gpurun_out/odd_trace_$w/t_results.db --by-grid | grep -E "^\| .k_(warp|unsharp_[ts]|pyr|coll|tile|upload|raster)" | awk -F'|' 'NF>7{printf "  %-28s grid %10s  %8s us\n", $2, $3, $6}'
done
