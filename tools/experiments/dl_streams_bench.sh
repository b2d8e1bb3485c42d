cd $GRAFT_REPO_ROOT
for mode in streams events streams events; do
  if [ $mode = streams ]; then export POPPY_HIP_DL_STREAMS=1; else unset POPPY_HIP_DL_STREAMS; fi
  timeout 900 python3 bench.py > gpurun_out/dlb_$mode.json 2> gpurun_out/dlb_$mode.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/dlb_$mode.json').read().strip().split('\n')[-1])
print('$mode', 'value', d.get('value'), 'unselected', d.get('value_unselected'), 'sequential', d.get('sequential_fps'), 'resident+writer', d.get('resident_pair_fps_with_writer'), 'cfg3', d['cfg3_4k']['value'], '480', d['scaling_baseline_480']['fps'], 'parity', d['parity_check']['equal'], d['cfg3_4k']['parity_check']['equal'], d['scaling_baseline_480'].get('parity_check',{}).get('equal'))
PY
done
