# EXPERIMENT (round 4): the pair set-up's streams confined to n compute units (hipExtStreamCreateWithCUMask) so that the other contexts' frames keep the rest
cd $GRAFT_REPO_ROOT
for p in "X=0" "POPPY_SETUP_CUS=128" "POPPY_SETUP_CUS=64" "POPPY_SETUP_CUS=96 POPPY_SETUP_CU_SPREAD=1" "POPPY_SETUP_CUS=64 POPPY_SETUP_CU_SPREAD=1" "X=0" "POPPY_SETUP_CUS=32 POPPY_SETUP_CU_SPREAD=1"; do
  echo "[$p] $(env $p timeout 300 python3 tools/experiments/setup_interference.py 2 1 3 2>&1 | tail -1) | pool $(env $p timeout 300 python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | alone $(env $p timeout 300 python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done
