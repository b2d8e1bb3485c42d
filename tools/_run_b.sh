python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/_exp2.py 2>&1 | tail -1
