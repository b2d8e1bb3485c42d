cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_sequences.py -x -q -m gpu 2>&1 | tail -2
for c in 6 4 8; do
timeout 900 python3 bench.py --contexts $c > gpurun_out/gb_$c.json 2> gpurun_out/gb_$c.err
python3 - <<PY
import json
d=json.loads(open('gpurun_out/gb_$c.json').read().strip().split('\n')[-1])
print('contexts $c', 'value', d.get('value'), 'unselected', d.get('value_unselected'), 'one call', d['pairs_in_one_call']['fps'], 'sequential', d.get('sequential_fps'), '480', d['scaling_baseline_480']['fps'], 'cfg3', d['cfg3_4k']['value'], 'step_ms', d['step_ms'])
PY
done
