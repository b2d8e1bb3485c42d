# Round 5's fuzz battery (new seeds; second run on the final build: seeds 91-99): frames incl. odd widths; the whole set-up against the oracle as shipped, with every image's medians by column histograms
# (from ksize 25 and from ksize 9), with the many-valued images' large windows by column histograms (windows of ranks), with short ORB buffers
cd $GRAFT_REPO_ROOT
echo "A frames debug          $(python tools/experiments/fuzz_frames.py 1200 91 1.0 2>&1 | tail -1)"
echo "B frames fused          $(python tools/experiments/fuzz_frames.py 1200 92 1.0 nodebug 2>&1 | tail -1)"
echo "C frames fused x2.5     $(python tools/experiments/fuzz_frames.py 300 93 2.5 nodebug 2>&1 | tail -1)"
echo "E set-up as shipped     $(python tools/experiments/fuzz_setup.py 50 94 2>&1 | tail -1)"
echo "F set-up cols forced    $(POPPY_MED_COLS_FORCE=1 python tools/experiments/fuzz_setup.py 30 95 2>&1 | tail -1)"
echo "G set-up cols from 9    $(POPPY_MED_COLS_FORCE=1 POPPY_MED_COLS_MIN=9 python tools/experiments/fuzz_setup.py 30 96 2>&1 | tail -1)"
echo "H set-up hard from 33   $(POPPY_MED_COLS_MIN_HARD=33 python tools/experiments/fuzz_setup.py 30 97 2>&1 | tail -1)"
echo "I set-up short buffers  $(POPPY_ORB_CAP=40 POPPY_ORB_KPCAP=32 python tools/experiments/fuzz_setup.py 25 98 2>&1 | tail -1)"
echo "J set-up upload both, gabor2 at 0   $(POPPY_SETUP_UPLOAD_BOTH=1 POPPY_GABOR2_AT=0 python tools/experiments/fuzz_setup.py 20 99 2>&1 | tail -1)"
echo "K set-up, chains one after the other   $(POPPY_SETUP_SERIAL=1 python tools/experiments/fuzz_setup.py 30 90 2>&1 | tail -1)"
