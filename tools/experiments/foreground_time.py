import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repo root
from poppy_amd import capi, synth
ctx = capi.Context(0)
for (w,h) in [(1920,1080),(3840,2160)]:
    img = synth.gen(w,h,1234)
    ctx.foreground(img)
    t0=time.perf_counter(); n=5
    for _ in range(n): ctx.foreground(img)
    dt=(time.perf_counter()-t0)/n
    print(f"{w}x{h}: foreground {dt*1e3:.2f} ms per image (incl. H2D of the image and D2H of the result)")
