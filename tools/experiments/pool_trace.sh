cd $GRAFT_REPO_ROOT
rm -f /tmp/pt_a.txt /tmp/pt_b.txt
POPPY_POOL_TRACE=/tmp/pt_a.txt python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1
python3 tools/experiments/pool_trace.py /tmp/pt_a.txt
POPPY_POOL_LANES=1 POPPY_POOL_TRACE=/tmp/pt_b.txt python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1
python3 tools/experiments/pool_trace.py /tmp/pt_b.txt
cp /tmp/pt_a.txt gpurun_out/pool_trace_default.txt
