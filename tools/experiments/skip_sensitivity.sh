# EXPERIMENT (timing build, results wrong): how much the pool gains if one of a frame's full-resolution kernels cost nothing (its grid cut to one workgroup)
cd $GRAFT_REPO_ROOT
for p in "X=0" "POPPY_X_SKIP=unsharp" "POPPY_X_SKIP=collapse0" "POPPY_X_SKIP=warp" "POPPY_X_SKIP=unsharp,collapse0" "POPPY_X_SKIP=unsharp,collapse0,warp" "X=0"; do
  echo "[$p] pool e2e $(env $p timeout 300 python3 tools/experiments/pool_e2e.py 30 3 6 2>&1 | tail -1 | grep -o '[0-9.]* frames/s') | resident x3 $(env $p timeout 300 python3 tools/experiments/concurrent_resident.py 3 20 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-200)"
done
