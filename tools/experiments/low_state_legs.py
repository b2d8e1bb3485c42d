#!/usr/bin/env python3
"""The pooled headline's low state inside ONE process: many legs (a new pool each: one allocating batch, three warm-up steps, ten timed steps), ms per step of each leg and of each of
its steps; with the stage stamps of one set-up per leg when POPPY_SETUP_TIMING is set.   usage: low_state_legs.py [legs] [torch|notorch] [contexts] [pairs]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
legs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
with_torch = (sys.argv[2] if len(sys.argv) > 2 else "torch") == "torch"
contexts = int(sys.argv[3]) if len(sys.argv) > 3 else 3
PAIRS = int(sys.argv[4]) if len(sys.argv) > 4 else 6
if with_torch:
    import torch
    torch.cuda.init()
from poppy_amd import capi, synth
W, H, FRAMES = 1920, 1080, 60
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
extra = capi.Context(0, number_of_frames=FRAMES) if os.environ.get("LEGS_EXTRA_CTX") else None      # (bench.py keeps one context alive beside its pools)
tuned = os.environ.get("LEGS_TUNED")                                                              # every second leg on a pool from poppy_hip_pool_create_tuned
for leg in range(legs):
    pool = capi.Pool([0], contexts_per_device=contexts, number_of_frames=FRAMES, **({"tuned_for": (W, H)} if tuned and leg % 2 else {}))
    for _ in range(4):
        pool.morph_pairs_device_counted(ptrs, W, H, -1.0)
    hip.hipDeviceSynchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); pool.morph_pairs_device_counted(ptrs, W, H, -1.0); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"leg {leg}: {sum(ts) / len(ts):6.2f} ms per step (min {min(ts):.2f}, max {max(ts):.2f}) -> {PAIRS * FRAMES / (sum(ts) / len(ts)) * 1e3:.0f} frames/s", flush=True)
    pool.close()
    if os.environ.get("LEGS_SLEEP"): time.sleep(float(os.environ["LEGS_SLEEP"]))
