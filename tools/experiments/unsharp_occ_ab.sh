#!/bin/bash
# Round 6: k_unsharp_stream with an LDS ring of 6 rows (20.7 KB per workgroup, six waves per SIMD, 80 VGPRs + 9 spilled) against 8 rows (27.6 KB, five waves, 78 VGPRs):
# altA.so = 8 rows, altB.so = 6 rows; rows per segment 0 (default: ~5120 waves) / 22 (~6144 waves).   gpurun -- bash tools/experiments/unsharp_occ_ab.sh
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for rep in 1 2; do for v in A B; do for rows in 0 22 18; do
  [ $rows = 0 ] && unset POPPY_UNSHARP_ROWS || export POPPY_UNSHARP_ROWS=$rows
  export POPPY_HIP_LIB=$R/poppy_amd/alt$v.so
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab -o t -- python3 $R/tools/experiments/frames_only.py 3840 2160 60 chain 3 > /dev/null 2>&1
  echo "build $v rows $rows: $(python3 $R/tools/rocprof_summary.py $O/ab/*.db 2>/dev/null | grep -E "k_unsharp_stream" | head -1)"; rm -rf $O/ab
done; done; done
