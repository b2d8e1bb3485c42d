python -m pytest tests -m gpu -x -q 2>&1 | tail -2
POPPY_HIP_HOSTPROF=1 python tools/_exp.py chain 0 2>&1 | tail -2
POPPY_HIP_HOSTPROF=1 python tools/_exp.py phase 0 2>&1 | tail -2
POPPY_HIP_SLOTS=6 python tools/_exp.py phase 0 2>&1 | tail -1
POPPY_HIP_SLOTS=8 python tools/_exp.py phase 0 2>&1 | tail -1
