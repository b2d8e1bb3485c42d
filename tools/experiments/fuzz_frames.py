"""Differential fuzz of the per-frame path: random sizes (all widths, so both warp kernels run), random point sets (with
duplicates, points on the border, strong deformations), random ratios; every frame and its triangle map / warped sources
against the oracle, bit for bit.   python tools/experiments/fuzz_frames.py [cases] [seed] [size scale] [nodebug]
(nodebug: a context outside debug mode - raster fused into the warp kernel, riding completion events - no triangle map to compare)"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as O
from poppy_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
nodebug = len(sys.argv) > 4 and sys.argv[4] == 'nodebug'
ctx = capi.Context(0); ctx.set_debug(not nodebug)
bad = 0; kinds = {0: 0, 1: 0, 2: 0}; t0 = time.time()
for i in range(cases):
    w = int(rng.integers(8, int(420 * scale))); h = int(rng.integers(6, int(300 * scale)))
    if rng.random() < 0.5: w = max(8, w & ~3)
    n = int(rng.integers(3, int(90 * scale)))
    p1 = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
    mode = rng.integers(0, 4)
    if mode == 0:   p2 = p1 + rng.normal(0, 3.0, (n, 2))
    elif mode == 1: p2 = p1 + rng.normal(0, 0.25 * min(w, h), (n, 2))                       # wild
    elif mode == 2:                                                                         # rotation + scale
        a = rng.uniform(-1.0, 1.0); sc = rng.uniform(0.5, 1.6); c = np.array([(w - 1) / 2, (h - 1) / 2])
        R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) * sc
        p2 = (p1 - c) @ R.T + c
    else:           p2 = np.round(p1 + rng.normal(0, 2.0, (n, 2)))                          # integer grid: ties, duplicates
    p2 = p2.astype(np.float32)
    p2[:, 0] = np.clip(p2[:, 0], 0, w - 1); p2[:, 1] = np.clip(p2[:, 1], 0, h - 1)
    if rng.random() < 0.7:
        cn = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]], np.float32)
        p1 = np.concatenate([p1, cn]); p2 = np.concatenate([p2, cn])
    if rng.random() < 0.3: p1[rng.integers(0, len(p1))] = p1[0]                             # a duplicate point
    c1 = synth.textured_bgr(w, h, int(rng.integers(1, 1000))); c2 = synth.textured_bgr(w, h, int(rng.integers(1, 1000)))
    g = synth.unit_field(w, h, int(rng.integers(1, 1000)))
    s = float(rng.uniform(0.02, 0.98)); m = float(rng.uniform(0.02, 0.98))
    try:
        want, wmp, d = O.morph_images(c1, c2, g, p1, p2, s, m, 64, debug=True)
    except Exception as e:
        # the oracle refuses what the reference would throw on (point on the far border): the library must refuse too
        try:
            ctx.morph_images(c1, c2, g, p1, p2, s, m); print(f"case {i}: oracle raised ({e}) but the library rendered"); bad += 1
        except capi.PoppyError:
            pass
        continue
    try:
        got, gmp = ctx.morph_images(c1, c2, g, p1, p2, s, m)
    except capi.PoppyError as e:
        print(f"case {i} {w}x{h} n={len(p1)} mode={mode}: library raised {e}"); bad += 1; continue
    kinds[ctx.last_warp_kind()] += 1
    checks = [("frame", got, want), ("points", gmp, wmp)]
    checks += [("trImg1", ctx.fetch("trImg1"), d["trImg1"]), ("trImg2", ctx.fetch("trImg2"), d["trImg2"])]
    if not nodebug:
        checks += [("triMap", ctx.fetch("triMap"), d["triMap"])]
    for name, a, b in checks:
        av = a.view(np.uint32) if a.dtype == np.float32 else a
        bv = b.view(np.uint32) if b.dtype == np.float32 else b
        if a.shape != b.shape or (av != bv).any():
            print(f"case {i} {w}x{h} n={len(p1)} mode={mode} s={s:.3f}: {name} differs in {(av != bv).sum() if a.shape == b.shape else 'shape'}"); bad += 1; break
print(f"{cases} cases, {bad} mismatches, kernels: general {kinds[0]}, tiled on an id map {kinds[1]}, fused raster {kinds[2]}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
