#!/bin/bash
# Timing-only ablation builds of the fused warp kernel (CONTAINER: hipcc cross-compiles): poppy_amd/abl_<mask>.so = the library with kernels_warp_bin.hip
# compiled with -DPOPPY_WARP_ABL=<mask> (warp_fast_device.h: 1 no division, 2 constant weights, 4 no blend arithmetic, 8 no footprint loads).  Results of such
# builds are wrong by construction; tools/experiments/abl_run.sh times them on one box.  The files are scratch (git-ignored, travel with gpurun).
set -euo pipefail
cd "$(dirname "$0")/../.."
python3 -c "import poppy_amd.build as b; b.build()" > /dev/null
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude"
for m in ${@:-0 1 2 4 8 15}; do
  /opt/rocm/bin/hipcc $FL -DPOPPY_WARP_ABL=$m -x hip -c poppy_amd/csrc/kernels_warp_bin.hip -o /tmp/abl_wb_$m.o
  objs=$(ls poppy_amd/build/*.o | grep -v kernels_warp_bin.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o poppy_amd/abl_$m.so $objs /tmp/abl_wb_$m.o -ldl -lpthread
  echo "poppy_amd/abl_$m.so"
done
