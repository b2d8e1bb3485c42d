cd $GRAFT_REPO_ROOT
for r in 2 3 4 6; do
echo "ring $r: pool $(POOL_WITH_TORCH=1 POPPY_HIP_RING=$r POPPY_HIP_SLOTS=$((r>4?r+1:4)) timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1) | 6x6 $(POOL_WITH_TORCH=1 POPPY_HIP_RING=$r POPPY_HIP_SLOTS=$((r>4?r+1:4)) timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1) | one ctx $(POOL_WITH_TORCH=1 POPPY_HIP_RING=$r POPPY_HIP_SLOTS=$((r>4?r+1:4)) timeout 300 python3 tools/experiments/pool_nowriter.py 4 1 6 1 6 | tail -1)"
done
echo "events: pool $(POOL_WITH_TORCH=1 POPPY_HIP_DL_EVENTS=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1) | 6x6 $(POOL_WITH_TORCH=1 POPPY_HIP_DL_EVENTS=1 timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1)"
