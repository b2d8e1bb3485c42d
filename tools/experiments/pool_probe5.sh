cd $GRAFT_REPO_ROOT
export POPPY_HIP_LIB=$PWD/poppy_amd/libpoppy_hip_experiments.so
for rep in 1 2; do
echo "writer:              $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "writer, no bytes:    $(POPPY_DL_SKIP_COPY=1 timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 1 6 | tail -1)"
echo "no writer:           $(timeout 300 python3 tools/experiments/pool_nowriter.py 2 6 36 0 6 | tail -1)"
done
