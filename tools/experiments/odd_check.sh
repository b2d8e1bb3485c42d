cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_odd_widths.py tests/test_gpu_fused_warp.py tests/test_gpu_bstage.py -x -q -m gpu 2>&1 | tail -3
bash tools/experiments/odd_trace2.sh
cd /tmp; for sz in "1900 1080" "1918 1080" "1920 1080" "2000 1200"; do python3 $GRAFT_REPO_ROOT/tools/experiments/frames_only.py $sz 60 chain 5 | tail -1; done
