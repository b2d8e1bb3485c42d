#!/bin/bash
# Timing-only ablation builds of the library (poppy_amd/abl_<n>.so, results wrong by construction) on ONE box:
# per build the rocprofv3 kernel-trace average of the kernels matching the filter, chained frame loop at 1080p and 4K.
# Usage: gpurun -- bash tools/experiments/abl_run.sh <kernel name filter>
filt=${1:-k_warp_bin}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cp $R/poppy_amd/libpoppy_hip.so /tmp/orig.so
for so in $(ls $R/poppy_amd/abl_*.so | sort -t_ -k3 -n); do v=$(basename $so .so)
  cp $so $R/poppy_amd/libpoppy_hip.so
  for sz in "1920 1080" "3840 2160"; do set -- $sz
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/abl -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 2 > /dev/null 2>&1
    echo "$v $1: $(python3 $R/tools/rocprof_summary.py $O/abl/*.db 2>/dev/null | grep -E "$filt" | head -2 | tr '\n' ' ')"; rm -rf $O/abl
  done
done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
