python bench.py --steps 5 --warmup 2 > gpurun_out/b10.json; python -c "
import json; d=json.load(open('gpurun_out/b10.json')); print(d['value'], d['phase_mode_fps'], d['pair_setup_ms'], d['fps_including_setup'], d['roofline']['frac'], d['cpu_baseline']['value'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
