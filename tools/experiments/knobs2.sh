# with the frame copies off the compute queues' critical path: do the runtime's knobs matter now?  (pool of six, writer, runtime bundled with torch)
cd $GRAFT_REPO_ROOT
export POOL_WITH_TORCH=1
run() { echo "$1: 36 at once $(env $2 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1 | sed 's/.*: //') | 6x6 $(env $2 timeout 300 python3 tools/experiments/pool_nowriter.py 6 6 6 1 | tail -1 | sed 's/.*: //')"; }
run "default            " X=1
run "HSA_ENABLE_SDMA=1  " HSA_ENABLE_SDMA=1
run "HSA_ENABLE_SDMA=0  " HSA_ENABLE_SDMA=0
run "GPU_MAX_HW_QUEUES=6" GPU_MAX_HW_QUEUES=6
run "GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
run "GPU_MAX_HW_QUEUES=12" GPU_MAX_HW_QUEUES=12
run "queues 8 + sdma 1  " "GPU_MAX_HW_QUEUES=8 HSA_ENABLE_SDMA=1"
run "default again      " X=1
