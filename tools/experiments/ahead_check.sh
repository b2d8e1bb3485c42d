# prepare-ahead (the next chained frame's plan upload + expansion behind the current frame's launches) against POPPY_HIP_NO_PREPARE_AHEAD=1
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_gpu_bstage.py tests/test_gpu_sequences.py tests/test_gpu_odd_widths.py tests/test_gpu_fused_warp.py -x -q -m gpu > gpurun_out/ahead_tests.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/ahead_tests.txt
for rep in 1 2; do
for sz in "1920 1080" "3840 2160"; do
  echo "ahead    $(timeout 300 python3 tools/experiments/frames_only.py $sz 60 chain 5 | tail -1)"
  echo "no ahead $(POPPY_HIP_NO_PREPARE_AHEAD=1 timeout 300 python3 tools/experiments/frames_only.py $sz 60 chain 5 | tail -1)"
done
echo "ahead    pool, writer:    $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "no ahead pool, writer:    $(POPPY_HIP_NO_PREPARE_AHEAD=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "ahead    pool, no writer: $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 0 6 | tail -1)"
echo "no ahead pool, no writer: $(POPPY_HIP_NO_PREPARE_AHEAD=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 0 6 | tail -1)"
echo "ahead    one context, writer: $(timeout 300 python3 tools/experiments/pool_nowriter.py 4 1 6 1 6 | tail -1)"
echo "no ahead one context, writer: $(POPPY_HIP_NO_PREPARE_AHEAD=1 timeout 300 python3 tools/experiments/pool_nowriter.py 4 1 6 1 6 | tail -1)"
done
