# kernel times of the median forms per content and window (one trace per content so that the averages are per content)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/median_forms
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace -d $O/t -o t -- python3 $R/tools/experiments/median_forms.py ${1:-1920} ${2:-1080} ${3:-123} 2 > $O/log.txt 2>&1
python3 $R/tools/experiments/median_forms_table.py $O/t/*/*.db $O/log.txt > $O/table.txt 2>&1 || python3 $R/tools/experiments/median_forms_table.py $O/t/*.db $O/log.txt > $O/table.txt 2>&1
cat $O/table.txt
