# the GPU test-suite + the chained / phase loops at both sizes + a kernel trace of the chained 1080p loop (the usual check after a kernel or layout change)
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/quick_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/quick_tests.txt
for sz in "1920 1080" "3840 2160"; do for mode in chain phase; do timeout 300 python3 tools/experiments/frames_only.py $sz 60 $mode 5 | tail -1; done; done
R=$PWD; cd /tmp; export TMPDIR=/tmp
for sz in "1920 1080" "3840 2160"; do w=${sz% *}
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/quick_trace_$w -o t -- python3 $R/tools/experiments/frames_only.py $sz 60 chain 3 > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $R/gpurun_out/quick_trace_$w/t_results.db | grep -E "k_(warp|unsharp_[ts]|pyr|coll|tile|upload)" | awk -F'|' '{printf "  %-28s %8s\n", $2, $5}'
done
