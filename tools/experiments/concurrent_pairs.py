import sys, time, threading, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repo root
import bench
from poppy_amd import capi
K = int(sys.argv[1])
a,b,g,p1,p2 = bench.synth_inputs()
shapes = np.array([capi.lib().poppy_frame_ratio(j, 60, -1.0) for j in range(60)])
ctxs=[]
for k in range(K):
    c = capi.Context(0, number_of_frames=60); c.pair_load(a,b,g,p1,p2); ctxs.append(c)
def work(c, steps):
    for _ in range(steps):
        c.reset(); c.render_many(shapes, chain=True)
    c.sync()
for c in ctxs: work(c, 1)
steps=5
t0=time.perf_counter()
th=[threading.Thread(target=work, args=(c,steps)) for c in ctxs]
for t in th: t.start()
for t in th: t.join()
dt=time.perf_counter()-t0
print(f"{K} pairs concurrently, chained: {K*steps*60/dt:.1f} frames/s aggregate ({dt/(steps*60)*1e6:.1f} us per frame-slot)")
