cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
timeout 300 rocprofv3 --kernel-trace -d $O/ov -o t -- python3 $R/tools/experiments/setup_interference.py 2 1 1.0 > $O/ov.log 2>&1
tail -1 $O/ov.log
python3 $R/tools/experiments/overlap_stats.py $O/ov/*.db; rm -rf $O/ov
