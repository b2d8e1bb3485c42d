# what the parts of k_median_cols cost: launches with parts left out (POPPY_MED_COLS_SKIP bits: 1 warm-up, 2 runs, 4 row step, 8 delta build, 16 chain init)
# usage: median_skip.sh <forms> "<skip values>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/median_skip
rm -rf $O; mkdir -p $O
for skip in ${2:-0 8 4 2 29 31}; do
  POPPY_HIP_LIB=$R/poppy_amd/libpoppy_hip_experiments.so POPPY_MED_COLS_SKIP=$skip timeout 300 rocprofv3 --kernel-trace -d $O/s$skip -o t -- python3 $R/tools/experiments/median_forms.py 1920 1080 ${1:-3} 2 > $O/log$skip.txt 2>&1
  echo "== skip $skip"; python3 $R/tools/experiments/median_forms_table.py $O/s$skip/t_results.db $O/log$skip.txt | grep -v "^|--\|launches" | awk -F'|' '{print $2, $3, $4, $5, $6, $7, $8}' | tr '\n' ';'; echo
done
