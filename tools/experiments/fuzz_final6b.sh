# Round 6, the shipped library: a longer battery with new seeds (801-809)
cd $GRAFT_REPO_ROOT
echo "A frames debug            $(python3 tools/experiments/fuzz_frames.py 1500 801 1.0 2>&1 | tail -1)"
echo "B frames fused            $(python3 tools/experiments/fuzz_frames.py 3000 802 1.0 nodebug 2>&1 | tail -1)"
echo "C frames fused x2.5       $(python3 tools/experiments/fuzz_frames.py 1000 803 2.5 nodebug 2>&1 | tail -1)"
echo "D frames fused x4         $(python3 tools/experiments/fuzz_frames.py 150 804 4.0 nodebug 2>&1 | tail -1)"
echo "E fused x2.5, stream unsharp  $(POPPY_UNSHARP_STREAM=1 python3 tools/experiments/fuzz_frames.py 500 805 2.5 nodebug 2>&1 | tail -1)"
echo "G set-up as shipped       $(python3 tools/experiments/fuzz_setup.py 80 807 2>&1 | tail -1)"
echo "H set-up from device memory  $(FUZZ_DEVICE=1 python3 tools/experiments/fuzz_setup.py 60 808 2>&1 | tail -1)"
echo "S sequences, pools, queued batches  $(python3 tools/experiments/fuzz_sequences.py 400 809 2>&1 | tail -2 | tr '\n' ' ')"
