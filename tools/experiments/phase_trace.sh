cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for rs in "0 3" "0 1"; do set -- $rs
  timeout 300 rocprofv3 --kernel-trace -d $O/ph -o t -- python3 $R/tools/experiments/setup_interference.py $1 $2 1.0 > $O/ph.log 2>&1; tail -1 $O/ph.log | cut -c1-200
  python3 $R/tools/experiments/phase_concurrency.py $O/ph/*.db 2>&1 | cut -c1-260; rm -rf $O/ph
done
