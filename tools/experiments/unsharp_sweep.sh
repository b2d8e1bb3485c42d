cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
run() { # rows w h
  POPPY_UNSHARP_ROWS=$1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/ur -o t -- python3 $R/tools/experiments/frames_only.py $2 $3 60 chain 2 > $O/ur.log 2>&1
  echo "rows $1 $2: $(python3 $R/tools/rocprof_summary.py $O/ur/*.db 2>/dev/null | grep -E "unsharp_" | head -1)"; rm -rf $O/ur
}
for r in 9 10 11 12 13 14; do run $r 1920 1080; done
for r in 18 20 22 24 27 30 36; do run $r 3840 2160; done
POPPY_UNSHARP_TILE=1 run 0 1920 1080
POPPY_UNSHARP_TILE=1 run 0 3840 2160
