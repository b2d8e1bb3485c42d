cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_prefilter2.py -m gpu -x -q -k "gabor or content" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for k in photo textured synthetic; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/gx_$k -o t -- python3 $R/tools/experiments/setup_content.py $k 1920 1080 3 > $O/gx_$k.txt 2>&1
  python3 $R/tools/rocprof_summary.py $O/gx_$k/*.db > $O/gx_${k}_trace.md 2>/dev/null; rm -rf $O/gx_$k
  echo "$k: $(grep gabor_fft $O/gx_${k}_trace.md | cut -c1-60 | tr "\n" " ") $(grep "re-formed\|pair set-up" $O/gx_$k.txt | tr "\n" " ")"
done
