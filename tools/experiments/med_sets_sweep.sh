#!/bin/bash
# Median kernel times (both images of a pair side by side, as in the set-up) for several numbers of histogram sets per launch (POPPY_MED_SETS).
# Usage: gpurun -- bash tools/experiments/med_sets_sweep.sh "1024 768 512 384 256 192"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for n in ${1:-1024 512 256}; do
  POPPY_MED_SETS=$n timeout 600 rocprofv3 --kernel-trace --stats -d $O/medsets_$n -o t -- python3 $R/tools/experiments/pair_begin_time.py 6 > /dev/null 2>&1
  echo "sets $n: $(python3 $R/tools/rocprof_summary.py $O/medsets_$n/*.db 2>/dev/null | grep k_median | sed 's/| `k_median_u8<\([0-9]*\), \([0-9]\)>` | [0-9]* | [0-9.]* | \([0-9.]*\) |.*/\1:\3/' | sort -n | tr '\n' ' ')  set-up $(POPPY_MED_SETS=$n python3 $R/tools/experiments/pair_begin_time.py 20 | head -1 | sed 's/.*pair_begin \([0-9.]*\) ms.*/\1/') ms"
  rm -rf $O/medsets_$n
done
