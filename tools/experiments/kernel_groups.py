import sys, time, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repo root
import bench
from poppy_amd import capi
a,b,g,p1,p2 = bench.synth_inputs()
ctx = capi.Context(0, number_of_frames=60)
ctx.pair_load(a,b,g,p1,p2)
shapes = np.array([capi.lib().poppy_frame_ratio(j, 60, -1.0) for j in range(60)])
def step():
    ctx.reset(); ctx.render_many(shapes, chain=True)
for _ in range(3): step()
ctx.sync()
t0=time.perf_counter()
for _ in range(5): step()
ctx.sync(); dt=time.perf_counter()-t0
ctx.set_timing(1); step(); ctx.sync()
d={n:round(ms/cnt*1e3,1) for n,ms,cnt in ctx.timing_summary()}
print(f"px={os.environ.get('POPPY_TAIL_PX')} thr={os.environ.get('POPPY_TAIL_THREADS')}: {300/dt:.1f} fps {dt/300*1e6:.1f} us/frame; down+tail+up={d['pyrdown']+d['pyr_tail']+d['collapse']:.1f}", d)
