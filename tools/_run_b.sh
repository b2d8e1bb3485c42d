python -m pytest tests/test_gpu_prefilter.py -x -q 2>&1 | tail -2
python tools/_fgtime.py
