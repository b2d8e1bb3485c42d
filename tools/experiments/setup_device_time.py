#!/usr/bin/env python3
"""Median of N pair set-ups from DEVICE-resident raw pairs (the pool's / bench's path: poppy_hip_pair_begin_device), ms.  usage: setup_device_time.py [kind synthetic|photo|textured] [N]"""
import ctypes as C, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from poppy_amd import capi, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "synthetic"; N = int(sys.argv[2]) if len(sys.argv) > 2 else 25
W, H = 1920, 1080
if kind == "photo":
    import numpy as np
    z = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "photo_pair_720x405.npz"))
    a, b = synth.upscale_bgr(z["a"], W, H), synth.upscale_bgr(z["b"], W, H)
elif kind == "textured":
    a, b = synth.textured_bgr(W, H, 5), synth.textured_bgr(W, H, 6)
else:
    a, b = synth.gen_pair(W, H, seed=1234)
hip = C.CDLL("libamdhip64.so")
def dev(img):
    d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), C.c_size_t(img.nbytes)) == 0
    assert hip.hipMemcpy(d, img.ctypes.data_as(C.c_void_p), C.c_size_t(img.nbytes), 1) == 0; return d.value
da, db = dev(a), dev(b)
c = capi.Context(0, number_of_frames=60)
for _ in range(3): c.pair_begin_device(da, db, W, H)
ts = []
for _ in range(N):
    t0 = time.perf_counter(); c.pair_begin_device(da, db, W, H); ts.append((time.perf_counter() - t0) * 1e3)
print(f"{kind}: median {statistics.median(ts):.3f} ms, min {min(ts):.3f}")
