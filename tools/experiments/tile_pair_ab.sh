#!/bin/bash
# Round 6: k_tile_expand_pair (two tiles per workgroup, the second tile's chain of loads trailing the first's) against k_tile_expand: exactness on the fixtures with the
# pair form forced for every geometry, then kernel-trace averages at 4K and 1080p, chained and phase mode.   gpurun -- bash tools/experiments/tile_pair_ab.sh
# (needs profiles/r06_tile_pair.patch applied: the kernel and its POPPY_TILE_EXPAND_PAIR switch were removed after this measurement)
cd "$GRAFT_REPO_ROOT"
echo "forced pair form: $(POPPY_TILE_EXPAND_PAIR=1 python3 -m pytest tests/test_gpu_fused_warp.py tests/test_gpu_odd_widths.py tests/test_gpu_bstage.py tests/test_gpu_sequences.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|rror' | tail -1)"
echo "default:          $(python3 -m pytest tests/test_gpu_fused_warp.py tests/test_gpu_odd_widths.py -x -q -m gpu 2>&1 | grep -E 'passed|failed|rror' | tail -1)"
echo "fuzz forced pair: $(POPPY_TILE_EXPAND_PAIR=1 python3 tools/experiments/fuzz_frames.py 400 811 2.5 nodebug 2>&1 | tail -1)"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
for rep in 1 2; do for v in 0 1; do for sz in "3840 2160" "1920 1080"; do set -- $sz
  export POPPY_TILE_EXPAND_PAIR=$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab -o t -- python3 $R/tools/experiments/frames_only.py $1 $2 60 chain 3 > /dev/null 2>&1
  echo "pair=$v $1 chain: $(python3 $R/tools/rocprof_summary.py $O/ab/*.db 2>/dev/null | grep -E "k_tile_expand" | head -1)"; rm -rf $O/ab
done; done; done
unset POPPY_TILE_EXPAND_PAIR
for v in 0 1 0 1; do echo "pair=$v 4K phase mode: $(POPPY_TILE_EXPAND_PAIR=$v python3 $R/tools/experiments/frames_only.py 3840 2160 60 phase 3 2>&1 | tail -1)"; done
