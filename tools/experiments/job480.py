import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from poppy_amd import capi, synth, sharding
W, H = 1920, 1080
a, b = synth.gen_pair(W, H)
c = capi.Context(0, number_of_frames=1)
ts = sharding.phase_share(0, 1, 480)
c.pair_begin(a, b)
c.render_phases(ts, counted=True)
t0 = time.perf_counter()
for _ in range(3):
    c.pair_begin(a, b)
    n = c.render_phases(ts, counted=True)
dt = (time.perf_counter() - t0) / 3
print(f"480-frame job: {dt*1e3:.1f} ms, {480/dt:.0f} frames/s")
