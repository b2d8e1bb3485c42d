import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from poppy_amd import capi, synth
w, h = 3840, 2160
a, b = synth.gen_pair(w, h)
c = capi.Context(0, number_of_frames=1)
c.pair_begin(a, b)
for _ in range(2):
    t0 = time.perf_counter(); nf, d = c.pair_begin(a, b); print("pair_begin %.1f ms nfeatures %d" % ((time.perf_counter() - t0) * 1e3, nf))
