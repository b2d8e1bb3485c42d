# round 4: the pool and the set-up interference under more hardware queues (tools/micro/queue_map.hip: by default a process's streams share FOUR, and a stream waits behind whatever runs on the streams it shares one with)
cd $GRAFT_REPO_ROOT
for p in "X=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=24" "X=0" "GPU_MAX_HW_QUEUES=16"; do
  echo "[$p] pool e2e $(env $p timeout 300 python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | $(env $p timeout 300 python3 tools/experiments/setup_interference.py 2 1 3 2>&1 | tail -1) | alone $(env $p timeout 300 python3 tools/experiments/setup_content.py synthetic 1920 1080 7 2>&1 | tail -1 | grep -o 'pair set-up [0-9.]* ms') | bench $(env $p timeout 300 python3 bench.py --headline-only --steps 20 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
done
