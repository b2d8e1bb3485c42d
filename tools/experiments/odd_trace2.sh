cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for sz in "3836 2160" "3838 2160" "3839 2160" "3840 2160" "3842 2160" "3844 2160"; do w=${sz% *}
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/odd_trace_$w -o t -- python3 $R/tools/experiments/frames_only.py $sz 60 chain 2 > /dev/null 2>&1
echo "$sz: $(python3 $R/tools/rocprof_summary.py $R/gpurun_out/odd_trace_$w/t_results.db | grep -E "k_warp_bin|k_collapse_level<true>|k_unsharp_stream|k_pyrdown_level<true>" | awk -F'|' '{printf "%s %s; ", $2, $5}')"
done
