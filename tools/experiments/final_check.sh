cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh r04_z > gpurun_out/profile_round.log 2>&1; tail -2 gpurun_out/profile_round.log
