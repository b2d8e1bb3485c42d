cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_prefilter2.py tests/test_gpu_prefilter.py tests/test_gpu_astage.py tests/test_gpu_sequences.py -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.txt | tail -3
echo "E $(python tools/experiments/fuzz_setup.py 60 58 2>&1 | tail -1)"
