cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for k in synthetic photo textured; do
  POPPY_SETUP_TIMING=1 python3 $R/tools/experiments/setup_content.py $k 1920 1080 3 > $O/content_$k.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ct_$k -o t -- python3 $R/tools/experiments/setup_content.py $k 1920 1080 3 > /dev/null 2>&1
  python3 $R/tools/rocprof_summary.py $O/ct_$k/*.db > $O/content_${k}_trace.md 2>/dev/null; rm -rf $O/ct_$k
done
tail -n 30 $O/content_photo.txt
