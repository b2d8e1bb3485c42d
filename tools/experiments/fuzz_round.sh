cd $GRAFT_REPO_ROOT
echo "A $(python tools/experiments/fuzz_frames.py 1500 51 1.0 2>&1 | tail -1)"
echo "B $(python tools/experiments/fuzz_frames.py 1500 52 1.0 nodebug 2>&1 | tail -1)"
echo "C $(python tools/experiments/fuzz_frames.py 400 53 2.5 nodebug 2>&1 | tail -1)"
echo "D $(POPPY_UNSHARP_STREAM=1 python tools/experiments/fuzz_frames.py 800 54 1.0 nodebug 2>&1 | tail -1)"
echo "E $(python tools/experiments/fuzz_setup.py 60 55 2>&1 | tail -1)"
echo "F $(POPPY_ORB_CAP=40 POPPY_ORB_KPCAP=32 python tools/experiments/fuzz_setup.py 30 56 2>&1 | tail -1)"
