# the default sequence's plans started by the pair loader (round 6) against POPPY_HIP_NO_PLAN_AHEAD=1
cd $GRAFT_REPO_ROOT
timeout 1800 python3 -m pytest tests/test_gpu_sequences.py tests/test_gpu_bstage.py tests/test_gpu_astage.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
export POOL_WITH_TORCH=1
for rep in 1 2 3; do
echo "ahead:    one ctx $(timeout 300 python3 tools/experiments/pool_nowriter.py 4 1 6 1 6 | tail -1 | sed 's/.*: //') | 6x6 $(timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1 | sed 's/.*: //') | 36 at once $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1 | sed 's/.*: //')"
echo "no ahead: one ctx $(POPPY_HIP_NO_PLAN_AHEAD=1 timeout 300 python3 tools/experiments/pool_nowriter.py 4 1 6 1 6 | tail -1 | sed 's/.*: //') | 6x6 $(POPPY_HIP_NO_PLAN_AHEAD=1 timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1 | sed 's/.*: //') | 36 at once $(POPPY_HIP_NO_PLAN_AHEAD=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1 | sed 's/.*: //')"
done
POPPY_SEQ_TIMING=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 1 3 1 3 2>&1 | grep "sequence of" | tail -2 | cut -c1-200
