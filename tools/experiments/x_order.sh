cd $GRAFT_REPO_ROOT
for p in "X=0" "POPPY_X_ORDER=1" "X=0" "POPPY_X_ORDER=1"; do
  echo "[$p] $(env $p timeout 300 python3 tools/experiments/setup_interference.py 0 3 3 2>&1 | tail -1) | pool e2e $(env $p timeout 300 python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | bench (torch) $(env $p timeout 300 python3 bench.py --headline-only --steps 30 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
POPPY_X_ORDER=1 timeout 300 rocprofv3 --kernel-trace -d $O/ph -o t -- python3 $R/tools/experiments/setup_interference.py 0 3 1.0 > $O/ph.log 2>&1; tail -1 $O/ph.log | cut -c1-200
python3 $R/tools/experiments/phase_concurrency.py $O/ph/*.db 2>&1 | grep -v columns | cut -c1-200; rm -rf $O/ph
