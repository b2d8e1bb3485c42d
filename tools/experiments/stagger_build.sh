#!/bin/bash
# Timing builds with the wave-priority stagger (pyramid_device.h: POPPY_STAGGER) switched on per kernel (CONTAINER): poppy_amd/abl_s<mask>.so.
# Results are unchanged by these builds; tools/experiments/stagger_run.sh times them on one box.
set -euo pipefail
cd "$(dirname "$0")/../.."
python3 -c "import poppy_amd.build as b; b.build()" > /dev/null
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude"
FILES="kernels_warp_bin kernels_frame kernels_pyramid_vec kernels_pyramid_cone kernels_pyramid_fused kernels_unsharp_stream"
rm -f poppy_amd/abl_*.so
for m in ${@:-0 1 3 5 9 17 33 65 129 255}; do
  for f in $FILES; do
    extra=""; [ $f = kernels_unsharp_stream ] && extra="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc $FL $extra -DPOPPY_STAGGER=$m -x hip -c poppy_amd/csrc/$f.hip -o /tmp/abl_st_${m}_$f.o &
  done
  wait
  objs=$(ls poppy_amd/build/*.o | grep -v -E "kernels_(warp_bin|frame|pyramid_vec|pyramid_cone|pyramid_fused|unsharp_stream).hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o poppy_amd/abl_s$m.so $objs /tmp/abl_st_${m}_*.o -ldl -lpthread
  echo "poppy_amd/abl_s$m.so"
done
