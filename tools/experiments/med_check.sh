cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_prefilter2.py -m gpu -x -q -k "not alternative_forms" 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -6
for k in synthetic photo textured; do
  echo "runs : $(python3 tools/experiments/setup_content.py $k 1920 1080 7 2>&1 | tail -1)"
  echo "old  : $(POPPY_MED_RUNS=0 python3 tools/experiments/setup_content.py $k 1920 1080 7 2>&1 | tail -1)"
done
