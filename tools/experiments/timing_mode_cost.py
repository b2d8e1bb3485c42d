"""What the live roofline measurement costs the frame loop: 60 chained 1080p frames per step with the warp kernel's
dispatch timestamps on (bench.py's timed region, set_timing(2)) and off.   python tools/experiments/timing_mode_cost.py"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from poppy_amd import capi, synth
W, H, N = 1920, 1080, 60
c1 = synth.textured_bgr(W, H, 1); c2 = synth.textured_bgr(W, H, 2); g = synth.unit_field(W, H, 3)
p1, p2 = synth.point_pairs(W, H, 440, seed=1, dup=0, oob=0)
ctx = capi.Context(0, number_of_frames=N)
ctx.pair_load(c1, c2, g, p1, p2)
shapes = np.array([capi.lib().poppy_frame_ratio(j, N, -1.0) for j in range(N)])
def run(mode, steps=10):
    ctx.set_timing(mode)
    for _ in range(2):
        ctx.reset(); ctx.render_many(shapes, chain=True)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(steps):
        ctx.reset(); ctx.render_many(shapes, chain=True)
    ctx.sync(); dt = time.perf_counter() - t0
    if mode: ctx.timing_summary()
    return steps * N / dt
if len(sys.argv) > 1 and sys.argv[1] == "off":      # only the untimed loop (for a kernel trace of the plain frame loop)
    print("timing off: %.0f fps" % run(0)); sys.exit(0)
for rep in range(3):
    print("timing off: %.0f fps   warp timestamps on: %.0f fps" % (run(0), run(2)))
