#!/bin/bash
# Three contexts of one process render no faster than one (6.5-6.9k frames/s without a writer) where three processes reach 11k: which runtime knob, if any,
# changes that?  pool_nowriter.py steps contexts pairs writer, under a few HIP / HSA environment settings.
cd "$GRAFT_REPO_ROOT"
run() { echo "$1 : nowriter $(env $1 python3 tools/experiments/pool_nowriter.py 20 3 6 0 2>&1 | tail -1) | writer $(env $1 python3 tools/experiments/pool_nowriter.py 20 3 6 1 2>&1 | tail -1)"; }
run "X=0"
run "AMD_DIRECT_DISPATCH=0"
run "HSA_ENABLE_INTERRUPT=0"
run "GPU_MAX_HW_QUEUES=8"
run "HIP_FORCE_DEV_KERNARG=1"
run "HSA_ENABLE_SDMA=0"
run "AMD_DIRECT_DISPATCH=0 GPU_MAX_HW_QUEUES=12"
run "ROC_ACTIVE_WAIT_TIMEOUT=100000"
