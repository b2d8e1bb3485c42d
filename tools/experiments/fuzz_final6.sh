# Round 6, final build: the fuzz battery once more with new seeds (frames against the oracle on the fused and the debug path, the set-up against the oracle, the sequence
# machinery incl. the queued pool batches against single frames)
cd $GRAFT_REPO_ROOT
echo "A frames debug            $(python3 tools/experiments/fuzz_frames.py 500 701 1.0 2>&1 | tail -1)"
echo "B frames fused            $(python3 tools/experiments/fuzz_frames.py 900 702 1.0 nodebug 2>&1 | tail -1)"
echo "C frames fused x2.5       $(python3 tools/experiments/fuzz_frames.py 300 703 2.5 nodebug 2>&1 | tail -1)"
echo "E fused x2.5, stream unsharp  $(POPPY_UNSHARP_STREAM=1 python3 tools/experiments/fuzz_frames.py 200 705 2.5 nodebug 2>&1 | tail -1)"
echo "G set-up as shipped       $(python3 tools/experiments/fuzz_setup.py 30 707 2>&1 | tail -1)"
echo "H set-up from device memory  $(FUZZ_DEVICE=1 python3 tools/experiments/fuzz_setup.py 20 708 2>&1 | tail -1)"
echo "S sequences, pools, queued batches  $(python3 tools/experiments/fuzz_sequences.py 120 709 2>&1 | tail -2 | tr '\n' ' ')"
