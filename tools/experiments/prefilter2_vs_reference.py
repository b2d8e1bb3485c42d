import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repo root
from poppy_amd import capi
import golden_util as G
ctx = capi.Context(0)
for case in ["a_256x256_chain", "a_512x384_chain"]:
    gf1 = G.full(case, "goodFeatures1")
    r = ctx.orb_input(gf1)
    us, gb, g1 = G.full(case, "us1"), G.full(case, "gb1"), G.full(case, "g1")
    det = G.full(case, "detail")
    if us is not None:
        print(case, "us exact:", (r["us"].view(np.uint32) == us.view(np.uint32)).mean(), np.abs(r["us"] - us).max())
        print(case, "gb max abs diff:", np.abs(r["gb"] - gb).max(), "differing floats", (r["gb"].view(np.uint32) != gb.view(np.uint32)).sum())
    dg = np.abs(r["g"].astype(int) - g1.astype(int))
    print(case, "g1 differing px:", (dg > 0).mean(), "max", dg.max())
    print(case, "detail:", r["detail"], "ref", det[0], "rel", abs(r["detail"] - det[0]) / det[0])
    inp = G.astage_inputs(case)
    gab = ctx.gabor_field(inp["img2"])
    ref = G.full(case, "gabor2")
    if ref is not None: print(case, "gabor2 max abs diff:", np.abs(gab - ref).max(), "differing floats", (gab.view(np.uint32) != ref.view(np.uint32)).sum())
    gfgpu = ctx.foreground(inp["img1"])
    print(case, "foreground == goodFeatures1:", np.array_equal(gfgpu, gf1))
    c = capi.Context(0, number_of_frames=int(inp["cfg"][0]))
    nf, d = c.pair_begin(inp["img1"], inp["img2"])
    print(case, "nfeatures", nf, "ref", int(det[3]), "details", d, det[:2])
    p1, p2 = c.pair_points()
    r1 = G.full(case, "prepared1")
    s_ref = set(map(tuple, np.round(r1, 3))); s_got = set(map(tuple, np.round(p1, 3)))
    print(case, "points:", len(p1), "ref", len(r1), "common", len(s_ref & s_got))
    frames = c.morph_frames(-1.0)
    for j, f in enumerate(frames[:3]):
        ref = G.full(case, f"frame{j}")
        if ref is not None:
            dd = np.abs(f.astype(int) - ref.astype(int))
            print(case, f"frame{j}: differing {(dd>0).mean():.4f} max {dd.max()} mean {dd.mean():.4f}")
    c.close()
