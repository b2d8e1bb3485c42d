#!/bin/bash
# The fused warp kernel's forms on ONE box: (1) tools/experiments/warp_variants.py (byte comparison + back-to-back relaunch), (2) per variant the
# rocprofv3 kernel-trace average inside the chained frame loop at 1080p and 4K (POPPY_WARP_VARIANT), (3) optionally the frame tests under a variant.
# usage: gpurun -- bash tools/experiments/warp_variants.sh "0 0x34 0x24 0x15" [variant to run the tests under]
vars=${1:-"0 0x34 0x24 0x15"}
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
python3 $R/tools/experiments/warp_variants.py "$(echo $vars | sed 's/0x//g; s/ /,/g')" 100 2>&1 | tee $O/warp_variants.txt
for v in $vars; do
  for sz in "1920 1080" "3840 2160"; do set -- $sz
    POPPY_WARP_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/wv -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 3 > $O/wv.log 2>&1
    echo "variant $v $1: $(grep 'frames/s' $O/wv.log) | $(python3 $R/tools/rocprof_summary.py $O/wv/*.db 2>/dev/null | grep -E 'k_warp|k_tile_expand' | tr '\n' ' ')" | tee -a $O/warp_variants.txt
    rm -rf $O/wv
  done
done
if [ -n "$2" ]; then
  cd $R && POPPY_WARP_VARIANT=$2 timeout 900 python3 -m pytest tests/test_gpu_fused_warp.py tests/test_gpu_bstage.py -m gpu -x -q 2>&1 | tail -5 | tee -a $O/warp_variants.txt
fi
