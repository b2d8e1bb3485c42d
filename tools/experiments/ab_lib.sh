#!/bin/bash
# A/B(/C) of two or three builds of the library on ONE box (kernel times differ between boxes by more than most changes): poppy_amd/altA.so, altB.so
# Usage: gpurun -- bash tools/experiments/ab_lib.sh <kernel name filter>
filt=${1:-k_unsharp_tile}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cp $R/poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2; do for v in A B C; do [ -f $R/poppy_amd/alt$v.so ] || continue
  cp $R/poppy_amd/alt$v.so $R/poppy_amd/libpoppy_hip.so
  for sz in "1920 1080" "3840 2160"; do set -- $sz
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 3 > /dev/null 2>&1
    echo "build $v $1: $(python3 $R/tools/rocprof_summary.py $O/ab/*.db 2>/dev/null | grep -E "$filt" | head -2 | tr '\n' ' ')"; rm -rf $O/ab
  done
done; done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
