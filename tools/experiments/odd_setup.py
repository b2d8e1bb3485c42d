import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from poppy_amd import capi, synth
os.environ["POPPY_SETUP_TIMING"] = "1"
for (w, h) in [(640, 480), (639, 480), (1920, 1080), (1918, 1080)]:
    a, b = synth.gen_pair(w, h, seed=1234)
    c = capi.Context(0)
    c.pair_begin(a, b)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); c.pair_begin(a, b); t.append((time.perf_counter() - t0) * 1e3)
    print(w, h, "pair_begin ms", sorted(t)[2], flush=True)
    c.close()
