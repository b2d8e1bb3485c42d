python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/_exp2.py
POPPY_HIP_NOFUSE=1 python tools/_exp2.py
python tools/_exp2.py
