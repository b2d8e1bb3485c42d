# Round 6's fuzz battery (new seeds): frames through the cone kernel (every depth: small images give shallow cones), odd widths on the fused path (the warp kernel's
# unaligned-row instantiation, padded level 1, the streaming unsharp forced for every size), the priorities; the whole set-up against the oracle (staged upload + gabor2 hand-shake)
cd $GRAFT_REPO_ROOT
echo "A frames debug            $(python3 tools/experiments/fuzz_frames.py 800 601 1.0 2>&1 | tail -1)"
echo "B frames fused            $(python3 tools/experiments/fuzz_frames.py 1200 602 1.0 nodebug 2>&1 | tail -1)"
echo "C frames fused x2.5       $(python3 tools/experiments/fuzz_frames.py 400 603 2.5 nodebug 2>&1 | tail -1)"
echo "D frames fused x4         $(python3 tools/experiments/fuzz_frames.py 60 604 4.0 nodebug 2>&1 | tail -1)"
echo "E fused x2.5, stream unsharp  $(POPPY_UNSHARP_STREAM=1 python3 tools/experiments/fuzz_frames.py 300 605 2.5 nodebug 2>&1 | tail -1)"
echo "F fused x2.5, no cone     $(POPPY_HIP_NOCONE=1 python3 tools/experiments/fuzz_frames.py 150 606 2.5 nodebug 2>&1 | tail -1)"
echo "G set-up as shipped       $(python3 tools/experiments/fuzz_setup.py 40 607 2>&1 | tail -1)"
echo "H set-up from device memory  $(FUZZ_DEVICE=1 python3 tools/experiments/fuzz_setup.py 30 608 2>&1 | tail -1)"
