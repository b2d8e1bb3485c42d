#!/bin/bash
# A/B of two or three builds of the library on ONE box for the pair set-up: poppy_amd/altA.so, altB.so[, altC.so]; median of 30 set-ups each, three rounds.
# Usage: gpurun -- bash tools/experiments/ab_setup.sh
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cp $R/poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2 3; do for v in A B C; do [ -f $R/poppy_amd/alt$v.so ] || continue
  cp $R/poppy_amd/alt$v.so $R/poppy_amd/libpoppy_hip.so
  echo "build $v: $(python3 $R/tools/experiments/pair_begin_time.py 30 2>&1 | grep pair_begin)"
done; done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
