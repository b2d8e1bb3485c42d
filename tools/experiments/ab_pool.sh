#!/bin/bash
# A/B of two or three builds of the library on ONE box for the bench's timed region (pool of 3 contexts, 6 pairs per step) and the same on one
# context: poppy_amd/altA.so, altB.so[, altC.so].   Usage: gpurun -- bash tools/experiments/ab_pool.sh
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cp $R/poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2; do for v in A B C; do [ -f $R/poppy_amd/alt$v.so ] || continue
  cp $R/poppy_amd/alt$v.so $R/poppy_amd/libpoppy_hip.so
  echo "build $v: $(python3 $R/tools/experiments/pool_e2e.py 10 3 6 2>&1 | tail -1)"
  echo "build $v: $(python3 $R/tools/experiments/pool_e2e.py 10 1 6 2>&1 | tail -1)"
done; done
cp /tmp/orig.so $R/poppy_amd/libpoppy_hip.so
