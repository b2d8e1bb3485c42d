"""Differential fuzz of the whole pair set-up (foreground x2, dft_detail2, ORB inputs, ORB detect, matcher, gabor2) against the oracle on
small random images of ragged sizes: nfeatures, the prepared point pairs and gabor2, bit for bit.
   python tools/experiments/fuzz_setup.py [cases] [seed]        (the oracle's set-up costs seconds per case at these sizes)"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as O
from poppy_amd import capi, synth

import os, ctypes as C
on_device = os.environ.get("FUZZ_DEVICE") == "1"
hip = C.CDLL("libamdhip64.so") if on_device else None
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; t0 = time.time()
for i in range(cases):
    w = int(rng.integers(96, 420)); h = int(rng.integers(80, 200))       # (from 256 x 64 up, widths divisible by 4 take the fused accumulate kernel)
    if rng.random() < 0.5: w &= ~3
    s = int(rng.integers(1, 10000))
    kind = rng.random()
    if kind < 0.4:
        a, b = synth.gen_pair(w, h, seed=s)
    elif kind < 0.7:
        a, b = synth.textured_bgr(w, h, s), synth.textured_bgr(w, h, s + 1)
    else:                                                                  # a random crop of the reference's sample photographs at a random scale
        sc = float(rng.uniform(1.0, 3.0))
        pw, ph = int(w * sc) + 1, int(h * sc) + 1
        pa, pb = synth.photo_pair(pw, ph)
        x0, y0 = int(rng.integers(0, pw - w + 1)), int(rng.integers(0, ph - h + 1))
        a, b = np.ascontiguousarray(pa[y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(pb[y0:y0 + h, x0:x0 + w])
    c = capi.Context(0, number_of_frames=4)
    try:
        if on_device:                                                      # FUZZ_DEVICE=1: the pair from device memory (poppy_hip_pair_begin_device: the device decides the median kernel per image)
            ptrs = []
            for img in (a, b):
                d = C.c_void_p(); arr = np.ascontiguousarray(img)
                assert hip.hipMalloc(C.byref(d), C.c_size_t(arr.nbytes)) == 0 and hip.hipMemcpy(d, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0
                ptrs.append(d)
            try:
                c.pair_begin_device(ptrs[0].value, ptrs[1].value, w, h)
                nf = c.pair_begin_info()[0]
            finally:
                for d in ptrs: hip.hipFree(d)
        else:
            nf, det = c.pair_begin(a, b)
        p1, p2 = c.pair_points()
        g = c.fetch("gabor2")
    except capi.PoppyError as e:
        lerr = str(e); p1 = p2 = g = None; nf = -1
    try:
        st = O.pair_setup(a, b)
    except Exception as e:
        st = None; oerr = str(e)
    c.close()
    if st is None or p1 is None:
        ok = (st is None) == (p1 is None)
        print(f"case {i} {w}x{h}: oracle {'error: ' + oerr if st is None else 'ok'}, library {'error: ' + lerr if p1 is None else 'ok'}")
    else:
        ok = (nf == st["nfeatures"] and p1.shape == st["points1"].shape and np.array_equal(p1.view(np.uint32), st["points1"].view(np.uint32))
              and np.array_equal(p2.view(np.uint32), st["points2"].view(np.uint32)) and np.array_equal(g.view(np.uint32), st["gabor2"].view(np.uint32)))
        if not ok:
            print(f"case {i} {w}x{h} seed {s}: MISMATCH nfeatures {nf} vs {st['nfeatures']}, points {p1.shape} vs {st['points1'].shape}, "
                  f"gabor2 differing {(g.view(np.uint32) != st['gabor2'].view(np.uint32)).sum() if g.shape == st['gabor2'].shape else 'shape'}"
                  + (f" (largest difference {np.abs(g.view(np.int32).astype(np.int64) - st['gabor2'].view(np.int32).astype(np.int64)).max()} ulp; points equal: "
                     f"{np.array_equal(p1, st['points1']) and np.array_equal(p2, st['points2'])})" if g.shape == st['gabor2'].shape and p1.shape == st['points1'].shape else ""))
    bad += 0 if ok else 1
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
