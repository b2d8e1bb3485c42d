#!/bin/bash
# Two builds of the library on ONE box (poppy_amd/alt_*.so, made in the container beside libpoppy_hip.so): the pair set-up per content with each.
# Usage: gpurun -- bash tools/experiments/median_kernel_ab.sh
R="$GRAFT_REPO_ROOT"; cd $R
cp poppy_amd/libpoppy_hip.so /tmp/orig.so
for rep in 1 2; do
for so in /tmp/orig.so $(ls poppy_amd/alt_*.so); do
  cp $so poppy_amd/libpoppy_hip.so
  for force in -1 1; do
    if [ $force = -1 ]; then unset POPPY_MED_COLS_FORCE; else export POPPY_MED_COLS_FORCE=$force; fi
    echo "$(basename $so) force=$force: synthetic $(python3 tools/experiments/setup_content.py synthetic 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | photo $(python3 tools/experiments/setup_content.py photo 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | textured $(python3 tools/experiments/setup_content.py textured 1920 1080 15 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | 4K synthetic $(python3 tools/experiments/setup_content.py synthetic 3840 2160 9 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
  done
done; done
unset POPPY_MED_COLS_FORCE
cp /tmp/orig.so poppy_amd/libpoppy_hip.so
