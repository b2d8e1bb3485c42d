cd $GRAFT_REPO_ROOT/tools/micro
echo "== default"; timeout 120 ./qm 12
echo "== GPU_MAX_HW_QUEUES=12"; GPU_MAX_HW_QUEUES=12 timeout 120 ./qm 12
