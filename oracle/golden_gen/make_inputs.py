#!/usr/bin/env python3
"""Writes the raw input files of every golden case into oracle/_dumps/cases/<case>/in/.

Inputs come from poppy_amd/synth.py (integer-defined), so the tests can regenerate them
on any box; only the reference OUTPUTS are committed as fixtures (see pack.py).
The same case table is imported by the tests (tests/golden_cases.py re-exports CASES).
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from poppy_amd import synth  # noqa: E402

CASES_DIR = os.path.join(ROOT, "oracle", "_dumps", "cases")

_DT = {np.dtype(np.uint8): "u8", np.dtype(np.float32): "f32", np.dtype(np.int32): "i32", np.dtype(np.float64): "f64"}


def write_in(case, name, arr):
    arr = np.ascontiguousarray(arr)
    d = os.path.join(CASES_DIR, case, "in")
    os.makedirs(d, exist_ok=True)
    shp = "x".join(str(s) for s in arr.shape)
    arr.tofile(os.path.join(d, f"{name}.{_DT[arr.dtype]}.{shp}.bin"))


# ---- case tables (single source of truth for generator AND tests) --------------------------
DISSOLVE = {
    # name: (w, h, phases): img2*phase + img1*(1-phase), the no-match fallback expression (src/poppy.hpp:129); -1 is what the
    # default CLI mode passes, 2.5 saturates both ways
    "x_dissolve_200x150": (200, 150, [-1.0, 0.0, 0.3, 0.5, 1.0 / 3.0, 1.0, 2.5]),
}


def dissolve_inputs(name):
    w, h, ph = DISSOLVE[name]
    return dict(img1=synth.textured_bgr(w, h, 71), img2=synth.textured_bgr(w, h, 72), phases=np.array(ph, dtype=np.float64))


BSTAGE = {
    # name: (w, h, npts, ratios[(shape, mask)], levels)
    "b_64x48": (64, 48, 12, [(0.0, 0.0), (1 / 60, 1 / 60), (0.25, 0.25), (0.5, 0.5), (1.0, 1.0)], 64),
    "b_256x256": (256, 256, 60, [(1 / 60, 1 / 60), (0.5, 0.5), (0.9, 0.7)], 64),
    "b_509x381": (509, 381, 150, [(0.5, 0.5), (0.3, 0.3)], 64),
    "b_1920x1080": (1920, 1080, 440, [(0.5, 0.5)], 64),
    # shallow pyramids (--pyramid N): the coarsest level is far larger than one workgroup's LDS
    "b_640x480_lv4": (640, 480, 100, [(0.5, 0.5), (0.2, 0.8)], 4),
    "b_1920x1080_lv4": (1920, 1080, 440, [(0.4, 0.4)], 4),
    "b_320x200_lv1": (320, 200, 40, [(0.5, 0.5)], 1),
}
ORB = {
    # name: (w, h, [nfeatures])
    "o_256x256": (256, 256, [300, 516]),
    "o_640x480": (640, 480, [500]),
    "o_1920x1080": (1920, 1080, [516]),
}
MATCH = {
    # name: (w, h, n, tolerance, seed)
    "m_640x480": (640, 480, 200, 1.0, 31),
    "m_1920x1080": (1920, 1080, 513, 1.0, 32),
    "m_tol2": (800, 600, 120, 2.0, 33),
}
ASTAGE = {
    # name: (w, h, nframes, phase, levels)
    "a_256x256_chain": (256, 256, 6, -1.0, 64),
    "a_256x256_phase": (256, 256, 1, 0.5, 64),
    "a_512x384_chain": (512, 384, 4, -1.0, 64),
    "a_256x256_align": (256, 256, 3, -1.0, 64, 1),      # enable_auto_align (src/matcher.cpp:29-32)
    "a_384x288_align": (384, 288, 2, -1.0, 64, 1),
    # round 2: (w, h, nframes, phase, levels, align, extra single-frame phases, input variant)
    "a_512x512_chain30": (512, 512, 30, -1.0, 64, 0, ()),                 # BASELINE.json configs[0]: frames hash-only
    "a_1920x1080_chain60": (1920, 1080, 60, -1.0, 64, 0, (0.25, 0.5)),    # configs[1] from the raw pair + two phase-mode frames
    "a_3840x2160_phase": (3840, 2160, 1, 0.5, 64, 0, ()),                 # configs[2]: one 4K phase-mode frame
    "a_256x256_phase01": (256, 256, 3, 0.0, 64, 0, (1.0,)),               # phase == 0 / == 1 short-circuits (src/poppy.hpp:54-70)
    # round 4: Settings::enable_radial_mask (the CLI's --radial; src/extractor.cpp:178-197, src/draw.cpp:21-38): flags bit 1
    "a_256x256_radial": (256, 256, 3, -1.0, 64, 2, ()),
    "a_320x200_radial": (320, 200, 2, -1.0, 64, 2, (0.5,)),
    # round 4: content that is not flat shapes — the reference's own sample photographs (tests/golden/photo_pair_720x405.npz, synth.photo_pair) and hash-noise textures
    "a_320x180_photo": (320, 180, 3, -1.0, 64, 0, (0.5,), "photo"),
    "a_256x192_textured": (256, 192, 3, -1.0, 64, 0, (), "textured"),
    "a_1920x1080_photo60": (1920, 1080, 60, -1.0, 64, 0, (), "photo"),    # BASELINE.json configs[1] on the photographs: frames pinned by sha256
    # round 5: two of the reference's own demo pairs whose widths are no multiples of 4 (make_demos.sh:15,31; tests/golden/demo_pairs.npz, synth.demo_pair):
    # the whole of poppy::morph at the sizes that took the general kernels until then
    "a_639x480_numbers": (639, 480, 4, -1.0, 64, 0, (0.5,), "numbers"),
    "a_749x480_cars": (749, 480, 4, -1.0, 64, 0, (0.5,), "cars"),
    # (A featureless second image does NOT reach the linear-blend fallback of src/poppy.hpp:125-134: with empty point lists
    #  Matcher::find -> morph_distance -> cv::convexHull throws "total >= 0 && (depth == CV_32F || depth == CV_32S)" first —
    #  tried with the generator.  The fallback expression itself is pinned by x_dissolve_* below.)
}


FSTAGE = {
    # name: (w, h, seed): one BGR image through Extractor::foreground (src/extractor.cpp:136-229)
    "f_160x120": (160, 120, 41),
    "f_317x211": (317, 211, 42),
    "f_640x480": (640, 480, 43),
}


DETAIL = {
    # name: (w, h, seed): dft_detail2 of a grey image; sizes chosen to hit every radix path of cv::dft
    "d_256x256": (256, 256, 51),       # powers of two: radix 4 only
    "d_512x384": (512, 384, 52),       # 512 = 4^4 * 2 (radix 2 pass), 384 = 128 * 3
    "d_640x480": (640, 480, 53),       # 640 = 128 * 5, 480 = 32 * 5 * 3
    "d_317x211": (317, 211, 54),       # padded to 320 x 216 = (64 * 5) x (8 * 3^3)
    "d_100x75": (100, 75, 55),         # 100 = 4 * 5^2, 75 = 3 * 5^2 (no power of two: the odd permutation branch)
    "d_1920x1080": (1920, 1080, 56),
}


def detail_inputs(name):
    w, h, seed = DETAIL[name]
    return {"gray": synth.textured_gray(w, h, seed)}


MARGIN = {
    # name: (w, h, union_w, union_h, seed): blur_margin (src/util.cpp:574-602)
    "g_same_200x150": (200, 150, 200, 150, 61),        # equal sizes: one-pixel strips still get blurred
    "g_wide_200x150_in_260x150": (200, 150, 260, 150, 62),
    "g_both_181x97_in_240x160": (181, 97, 240, 160, 63),
}


def margin_inputs(name):
    w, h, uw, uh, seed = MARGIN[name]
    return {"img": synth.textured_bgr(w, h, seed), "cfg": np.array([uw, uh], dtype=np.float64)}


def fstage_inputs(name):
    w, h, seed = FSTAGE[name]
    return {"img1": synth.textured_bgr(w, h, seed)}


def bstage_inputs(name):
    w, h, n, ratios, levels = BSTAGE[name]
    c1 = synth.textured_bgr(w, h, 21)
    c2 = synth.textured_bgr(w, h, 22)
    gabor2 = synth.unit_field(w, h, 11)
    p1, p2 = synth.point_pairs(w, h, n, seed=5)
    return dict(c1=c1, c2=c2, gabor2=gabor2, pts1=p1, pts2=p2,
                ratios=np.array(ratios, dtype=np.float64).ravel(), levels=np.array([levels], dtype=np.float64))


def orb_inputs(name):
    w, h, nf = ORB[name]
    g1 = synth.textured_gray(w, h, 7)
    g2 = np.roll(synth.textured_gray(w, h, 7), (h // 100 + 1, w // 50 + 1), axis=(0, 1))
    return dict(g1=g1, g2=np.ascontiguousarray(g2), nfeatures=np.array(nf, dtype=np.float64))


def match_inputs(name):
    w, h, n, tol, seed = MATCH[name]
    p1, p2 = synth.point_pairs(w, h, n, seed=seed, dup=3, oob=2)
    p1, p2 = p1[:-4], p2[:-4]                       # no corners: Matcher::prepare adds them
    rng = synth.XorShift64Star(seed + 100)          # break the 1:1 order so greedy NN has work to do
    perm = np.arange(len(p2))
    for i in range(len(perm) - 1, 0, -1):
        j = rng.uniform(0, i + 1)
        perm[i], perm[j] = perm[j], perm[i]
    return dict(pts1=p1, pts2=np.ascontiguousarray(p2[perm]), cfg=np.array([w, h, tol], dtype=np.float64))


def astage_inputs(name):
    t = ASTAGE[name]
    w, h, nframes, phase, levels = t[:5]
    align = t[5] if len(t) > 5 else 0
    extra = list(t[6]) if len(t) > 6 else []
    variant = t[7] if len(t) > 7 else ""
    a, b = synth.gen_pair(w, h)
    if variant == "photo":
        a, b = synth.photo_pair(w, h)
    if variant == "textured":
        a, b = synth.textured_bgr(w, h, 7), synth.textured_bgr(w, h, 8)
    if variant in ("numbers", "cars"):
        a, b = synth.demo_pair(variant)
        assert a.shape == (h, w, 3) and b.shape == (h, w, 3)
    if variant == "flat2":                          # a featureless second image: ORB finds nothing, the point lists come back empty
        b = np.full_like(a, 77)                     # (the reference throws on it, see ASTAGE; kept for the library's own error test)
    cfg = [nframes, phase, levels] + ([align] if (align or len(t) > 6) else []) + extra
    return dict(img1=a, img2=b, cfg=np.array(cfg, dtype=np.float64))


def prims_inputs():
    rng = synth.XorShift64Star(77)
    w, h = 320, 200
    pts = np.array([[rng.uniform(0, 4 * w) / 4.0, rng.uniform(0, 4 * h) / 4.0] for _ in range(65)], dtype=np.float32)
    polys = np.array([[rng.uniform(-20, w + 20), rng.uniform(-20, h + 20), rng.uniform(-20, w + 20), rng.uniform(-20, h + 20),
                       rng.uniform(-20, w + 20), rng.uniform(-20, h + 20)] for _ in range(40)], dtype=np.int32)
    src = synth.textured_bgr(w, h, 41)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    mx = (xx * np.float32(1.03) + yy * np.float32(0.05) - np.float32(7.3)).astype(np.float32)
    my = (yy * np.float32(0.97) - xx * np.float32(0.02) + np.float32(4.6)).astype(np.float32)
    mx[0, 0] = np.float32(1e12); my[0, 1] = np.float32(-1e12); mx[1, 0] = np.float32(np.nan)   # x86 cvRound indefinite
    mats = np.array([[1 + (rng.uniform(0, 200) - 100) / 500.0, (rng.uniform(0, 200) - 100) / 700.0, (rng.uniform(0, 200) - 100) / 3.0,
                      (rng.uniform(0, 200) - 100) / 700.0, 1 + (rng.uniform(0, 200) - 100) / 500.0, (rng.uniform(0, 200) - 100) / 3.0,
                      (rng.uniform(0, 200) - 100) / 1e6, (rng.uniform(0, 200) - 100) / 1e6, 1.0] for _ in range(20)], dtype=np.float32)
    mats[3] = 0            # singular -> zero inverse
    return dict(subdiv_pts=pts, subdiv_rect=np.array([w, h], dtype=np.float64), polys=polys,
                remap_src=src, remap_mx=mx, remap_my=my, mats33=mats)


ALIGN = {
    # name: (w, h, n, angle_deg, shift_x, shift_y, scale, seed)   second point set = first one rotated / shifted / scaled + jitter
    "l_320x240": (320, 240, 60, 7.0, 6.0, -4.0, 1.03, 61),
    "l_640x480": (640, 480, 150, -11.0, -9.0, 5.0, 0.97, 62),
    "l_317x211": (317, 211, 40, 3.0, 2.0, 3.0, 1.0, 63),
}


def align_inputs(name):
    w, h, n, ang, sx, sy, sc, seed = ALIGN[name]
    img2 = synth.textured_bgr(w, h, seed)
    rng = synth.XorShift64Star(seed)
    p1 = np.array([[w / 8.0 + rng.uniform(0, 3 * w) / 4.0, h / 8.0 + rng.uniform(0, 3 * h) / 4.0] for _ in range(n)], dtype=np.float64)
    a = np.deg2rad(ang)
    c = np.array([w / 2.0, h / 2.0])
    R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) * sc
    jit = np.array([[(rng.uniform(0, 200) - 100) / 100.0, (rng.uniform(0, 200) - 100) / 100.0] for _ in range(n)])
    p2 = (p1 - c) @ R.T + c + np.array([sx, sy]) + jit
    aff = np.array([[0.98, 0.07, 3.25], [-0.05, 1.03, -2.5]], dtype=np.float64)
    return dict(img2=img2, pts1=p1.astype(np.float32), pts2=p2.astype(np.float32), cfg=np.array([w, h], dtype=np.float64), aff=aff)


def all_cases():
    out = []
    out += [("bstage", n, bstage_inputs) for n in BSTAGE]
    out += [("orb", n, orb_inputs) for n in ORB]
    out += [("match", n, match_inputs) for n in MATCH]
    out += [("astage", n, astage_inputs) for n in ASTAGE]
    out += [("prims", "p_prims", lambda _n: prims_inputs())]
    out += [("fstage", n, fstage_inputs) for n in FSTAGE]
    out += [("detail", n, detail_inputs) for n in DETAIL]
    out += [("margin", n, margin_inputs) for n in MARGIN]
    out += [("align", n, align_inputs) for n in ALIGN]
    out += [("dissolve", n, dissolve_inputs) for n in DISSOLVE]
    return out


if __name__ == "__main__":
    only = sys.argv[1:]
    for mode, name, fn in all_cases():
        if only and name not in only:
            continue
        for k, v in fn(name).items():
            write_in(name, k, v)
        os.makedirs(os.path.join(CASES_DIR, name, "out"), exist_ok=True)
        print(mode, name)          # run_all.sh reads these "mode case" lines
