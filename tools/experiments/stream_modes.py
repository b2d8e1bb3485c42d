import sys, time, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repo root
import bench
from poppy_amd import capi
mode = sys.argv[1]; timing = int(sys.argv[2])
a,b,g,p1,p2 = bench.synth_inputs()
ctx = capi.Context(0, number_of_frames=60)
ctx.pair_load(a,b,g,p1,p2)
if mode == 'chain':
    shapes = np.array([capi.lib().poppy_frame_ratio(j, 60, -1.0) for j in range(60)])
else:
    shapes = np.array([j/60.0 for j in range(60)])
def step():
    ctx.reset(); ctx.render_many(shapes, chain=(mode=='chain'))
for _ in range(2): step()
capi.lib().poppy_hip_set_timing(ctx.h, timing); ctx.sync()
t0=time.perf_counter()
for _ in range(5): step()
ctx.sync(); dt=time.perf_counter()-t0
print(f"slots={os.environ.get('POPPY_HIP_SLOTS')} nograph={os.environ.get('POPPY_HIP_NOGRAPH')} head={os.environ.get('POPPY_HIP_HEADMODE')} hwq={os.environ.get('GPU_MAX_HW_QUEUES')} {mode} timing={timing}: {300/dt:.1f} fps  {dt/300*1e6:.1f} us/frame")
if timing: print(ctx.timing_summary())
