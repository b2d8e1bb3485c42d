cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused_warp.py tests/test_gpu_bstage.py tests/test_gpu_sequences.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/experiments/fuzz_frames.py 600 41 1.0 nodebug 2>&1 | tail -2
python tools/experiments/fuzz_frames.py 400 42 2.5 2>&1 | tail -2
timeout 90 tools/micro/wa 2>&1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for sz in "1920 1080" "3840 2160"; do set -- $sz
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/wv -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 3 > $O/wv.log 2>&1
  echo "$1: $(grep 'frames/s' $O/wv.log) | $(python3 $R/tools/rocprof_summary.py $O/wv/*.db 2>/dev/null | grep -E 'k_warp|k_tile_expand' | tr '\n' ' ')"
  rm -rf $O/wv
done
