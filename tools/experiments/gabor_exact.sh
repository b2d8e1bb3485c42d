cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "E $(python tools/experiments/fuzz_setup.py 60 55 2>&1 | tail -1)"
echo "G $(python tools/experiments/fuzz_setup.py 40 57 2>&1 | tail -1)"
