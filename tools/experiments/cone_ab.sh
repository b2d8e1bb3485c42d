# round 6: the collapse cone (one launch from the tail's level up to level 1 / 2) against round 5's launches (POPPY_HIP_NOCONE=1)
# parity first (b-stage + sequences), then the chained and phase frame loops at 1080p and 4K both ways, then a kernel trace of the chained 1080p loop
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06_cone; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_bstage.py tests/test_gpu_sequences.py tests/test_gpu_odd_widths.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
tail -3 $O/tests.txt
for rep in 1 2; do
for sz in "1920 1080" "3840 2160"; do
  for mode in chain phase; do
    echo "cone    $(timeout 300 python3 tools/experiments/frames_only.py $sz 60 $mode 5 | tail -1)"
    echo "no cone $(POPPY_HIP_NOCONE=1 timeout 300 python3 tools/experiments/frames_only.py $sz 60 $mode 5 | tail -1)"
  done
done
done | tee $O/ab.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for sz in "1920 1080" "3840 2160"; do
  w=${sz% *}
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$w -o t -- python3 $R/tools/experiments/frames_only.py $sz 60 chain 3 > $R/$O/trace_$w.log 2>&1
  python3 $R/tools/rocprof_summary.py $R/$O/trace_$w/t_results.db --by-grid > $R/$O/trace_$w.md 2>/dev/null || true
  head -16 $R/$O/trace_$w.md
done
