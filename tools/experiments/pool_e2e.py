#!/usr/bin/env python3
"""The bench's timed region WITHOUT torch in the process: 4 synthetic 1080p pairs per step through the library's pool of 2 contexts
(pair set-up from the raw images resident in HBM + 60 chained frames + writer hand-off).  With torch imported first the process runs
on the HIP / HSA runtime bundled in the torch wheel, under which every frame download is a blit kernel on the CUs; without it the
image's runtime sends them to the SDMA engines.   python tools/experiments/pool_e2e.py [steps] [contexts]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from poppy_amd import capi, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
contexts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H, FRAMES = 1920, 1080, 60
PAIRS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
capi.lib()
hip = ctypes.CDLL("libamdhip64.so")
ptrs = []
for k in range(PAIRS):
    a, b = synth.gen_pair(W, H, seed=1234 + k)
    pp = []
    for img in (a, b):
        d = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(img.nbytes)) == 0
        assert hip.hipMemcpy(d, img.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(img.nbytes), 1) == 0
        pp.append(d.value)
    ptrs.append(tuple(pp))
pool = capi.Pool([0], contexts_per_device=contexts, number_of_frames=FRAMES)
for _ in range(3):
    pool.morph_pairs_device_counted(ptrs, W, H, -1.0)
hip.hipDeviceSynchronize()
t0 = time.perf_counter()
n = 0
for _ in range(steps):
    n += pool.morph_pairs_device_counted(ptrs, W, H, -1.0)
hip.hipDeviceSynchronize()
dt = time.perf_counter() - t0
print(f"pool of {contexts} contexts, no torch in the process: {n / dt:.1f} frames/s end to end ({dt / steps * 1e3:.2f} ms per step of {PAIRS} pairs)")
