cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "6 calls x 6 pairs:  $(timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1)"
echo "1 call x 36 pairs:  $(timeout 300 python3 tools/experiments/pool_nowriter.py 1 6 36 1 6 | tail -1)"
echo "2 calls x 36 pairs: $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "1 call x 36, 4 ctx: $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 4 36 1 6 | tail -1)"
echo "1 call x 36, 8 ctx: $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 8 36 1 6 | tail -1)"
echo "no writer 36:       $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 0 6 | tail -1)"
done
