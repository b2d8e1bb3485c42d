#!/bin/bash
# Timing builds of the fused warp kernel with wave priorities / start offsets (CONTAINER): poppy_amd/abl_p<n>.so = the library with kernels_warp_bin.hip compiled
# with -DPOPPY_WARP_PRIO=<n> (see the top of that file).  Results are unchanged by these builds; tools/experiments/abl_run.sh times them on one box.
set -euo pipefail
cd "$(dirname "$0")/../.."
python3 -c "import poppy_amd.build as b; b.build()" > /dev/null
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude"
rm -f poppy_amd/abl_*.so
for m in ${@:-0 1 2 3 4 5}; do
  /opt/rocm/bin/hipcc $FL -DPOPPY_WARP_PRIO=$m -x hip -c poppy_amd/csrc/kernels_warp_bin.hip -o /tmp/abl_wp_$m.o
  objs=$(ls poppy_amd/build/*.o | grep -v kernels_warp_bin.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o poppy_amd/abl_p$m.so $objs /tmp/abl_wp_$m.o -ldl -lpthread
  echo "poppy_amd/abl_p$m.so"
done
