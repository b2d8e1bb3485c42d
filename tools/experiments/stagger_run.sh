#!/bin/bash
# Every poppy_amd/abl_*.so in turn on ONE box: frames/s of the chained loop (untraced) and the per-kernel averages of a traced one, 1080p and 4K.
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cp $R/poppy_amd/libpoppy_hip.so /tmp/orig.so
for so in $(ls $R/poppy_amd/abl_*.so | sort -t_ -k3 -V); do v=$(basename $so .so)
  cp $so $R/poppy_amd/libpoppy_hip.so
  for sz in "1920 1080" "3840 2160"; do set -- $sz
    fps=$(python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 5 | tail -1)
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/abl -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 2 > /dev/null 2>&1
    echo "$v $1: $fps"
    python3 $R/tools/rocprof_summary.py $O/abl/*.db 2>/dev/null | grep -E "k_(warp_bin|unsharp|pyrdown|collapse|tile_expand|pyr_tail)" | awk -F'|' '{printf "    %s %s\n", $2, $5}'; rm -rf $O/abl
  done
done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
