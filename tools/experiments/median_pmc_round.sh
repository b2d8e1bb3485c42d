# pipe counters of the median kernels for profiles/r05_median_pmc.txt: k_median_cols on shapes (one count per lane) and on a photograph (windows of ranks), k_median_u8 on the photograph; ksize 49
cd $GRAFT_REPO_ROOT
{ echo "# k_median_cols, one count per lane (shapes, ksize 49, form 2): pipe counters per dispatch (rocprofv3 --pmc, separate passes; tools/experiments/median_pmc.sh 49 2 shapes)"
  bash tools/experiments/median_pmc.sh 49 2 shapes 2>&1 | tail -40; echo
  echo "# the same on a photograph (tiles by windows of ranks)"; bash tools/experiments/median_pmc.sh 49 2 photo 2>&1 | tail -40; echo
  echo "# k_median_u8 (form 1) on the photograph, ksize 49"; bash tools/experiments/median_pmc.sh 49 1 photo 2>&1 | tail -40; } > gpurun_out/r05_median_pmc.txt 2>&1
tail -5 gpurun_out/r05_median_pmc.txt
