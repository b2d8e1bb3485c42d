cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_odd_widths.py tests/test_gpu_fused_warp.py tests/test_gpu_bstage.py -x -q -m gpu 2>&1 | tail -3
bash tools/experiments/odd_trace.sh 2>&1 | grep -E "==|warp_bin|pyrdown<|pyrdown_level<true>"
