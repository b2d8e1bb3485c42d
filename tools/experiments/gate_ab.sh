# the pool's set-up gate (POPPY_POOL_SETUPS = set-ups side by side per device; 0 = no gate) x the chains of a set-up (POPPY_POOL_CHAINS: 0 side by side, 1 one after the other):
# six contexts, the bench's steps (6 calls of 6 pairs) and 36 pairs in one call, torch's runtime
cd $GRAFT_REPO_ROOT
export POOL_WITH_TORCH=1
for rep in 1 2; do
for cfg in "0 1" "1 1" "1 0" "2 1" "2 0" "3 0"; do set -- $cfg
  echo "gate $1 chains $2: 6x6 $(POPPY_POOL_SETUPS=$1 POPPY_POOL_CHAINS=$2 timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1 | sed 's/.*: //') | 36 at once $(POPPY_POOL_SETUPS=$1 POPPY_POOL_CHAINS=$2 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1 | sed 's/.*: //') | 4 ctx 36: $(POPPY_POOL_SETUPS=$1 POPPY_POOL_CHAINS=$2 timeout 300 python3 tools/experiments/pool_nowriter.py 2 4 36 1 6 | tail -1 | sed 's/.*: //')"
done; done
