# the lockstep set-up phase of the pool (three set-ups side by side: six median chains share the GPU) under fewer, longer median segments (less warm-up work in total)
cd $GRAFT_REPO_ROOT
for p in "X=0" "POPPY_MED_SETS=512" "POPPY_MED_SETS=256" "POPPY_MED_SETS=384" "X=0" "POPPY_MED_SETS=512"; do
  echo "[$p] pool e2e $(env $p timeout 300 python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | three set-ups side by side: $(env $p timeout 300 python3 tools/experiments/setup_interference.py 0 3 3 2>&1 | tail -1) | alone $(env $p timeout 300 python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms')"
done
