cd $GRAFT_REPO_ROOT
for s in 4 6 8; do for r in 3 6; do echo "slots $s ring $r: $(POPPY_HIP_SLOTS=$s POPPY_HIP_RING=$r timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1)"; done; done
echo "seq timing, one context:"; POPPY_SEQ_TIMING=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 1 2 1 2>&1 | tail -4
echo "seq timing, six contexts:"; POPPY_SEQ_TIMING=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 6 1 2>&1 | tail -8
