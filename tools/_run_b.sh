python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b11.json; python -c "
import json; d=json.load(open('gpurun_out/b11.json')); print(d['value'], d['phase_mode_fps'], d['batched_pairs_fps'], d['pair_setup_ms'], d['fps_including_setup'])"
