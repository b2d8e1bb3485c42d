cd $GRAFT_REPO_ROOT
echo "A $(python tools/experiments/fuzz_frames.py 1500 71 1.0 2>&1 | tail -1)"
echo "B $(python tools/experiments/fuzz_frames.py 1500 72 1.0 nodebug 2>&1 | tail -1)"
echo "C $(python tools/experiments/fuzz_frames.py 400 73 2.5 nodebug 2>&1 | tail -1)"
echo "E $(python tools/experiments/fuzz_setup.py 60 74 2>&1 | tail -1)"
echo "F $(POPPY_ORB_CAP=40 POPPY_ORB_KPCAP=32 python tools/experiments/fuzz_setup.py 30 75 2>&1 | tail -1)"
