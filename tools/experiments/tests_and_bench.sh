cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_a_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/r06_a_tests.txt
tail -5 gpurun_out/r06_a_tests.txt
timeout 900 python3 bench.py > gpurun_out/r06_a_bench.json 2> gpurun_out/r06_a_bench.err; tail -c 600 gpurun_out/r06_a_bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_a_bench.json').read().strip().split('\n')[-1])
for k in ('value','value_unselected','ms_per_step','sequential_fps','resident_pair_fps','pair_setup_ms'):
    print(k, d.get(k))
print('cfg3', d.get('cfg3_4k',{}).get('value'), 'roofline', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('achieved'))
print('480', d.get('scaling_baseline_480',{}).get('fps'), d.get('scaling_baseline_480',{}).get('parity_check'))
print('content', d.get('content_sensitivity'))
PY
