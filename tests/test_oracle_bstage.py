"""The oracle's per-frame path (morph_images) against the reference fixtures.

Every stage boundary is compared BIT-EXACT with the arrays captured from the compiled reference
(tests/golden/manifest.json -> provenance).  This is what pins oracle/ before it is used to judge the HIP path.
"""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

CASES = ["b_64x48", "b_256x256", "b_509x381"]


@pytest.mark.parametrize("case", CASES)
def test_frame_stages_bit_exact(case):
    inp = G.bstage_inputs(case)
    w, h, n, ratios, levels = G.make_inputs.BSTAGE[case]
    for k, (s, m) in enumerate(ratios):
        pf = f"f{k}_"
        out, mp, d = O.morph_images(inp["c1"], inp["c2"], inp["gabor2"], inp["pts1"], inp["pts2"], s, m, levels, debug=True)
        G.check(case, pf + "morphedPoints", mp)
        nt6 = G.entries(case)[pf + "triangleList"]["shape"][0]
        G.check(case, pf + "triangleList", d["tri6"][:nt6])
        G.check(case, pf + "triIdx", d["idx3"])
        for name in ("triMap", "H", "M1", "M2", "mapx1", "mapy1", "mapx2", "mapy2", "trImg1", "trImg2",
                     "lbmask", "lapBlend", "unsharp"):
            G.check(case, pf + name, d[name], what=f"ratio {s}")
        G.check(case, pf + "frame", out)


@pytest.mark.slow
def test_frame_1080p_bit_exact():
    case = "b_1920x1080"
    inp = G.bstage_inputs(case)
    w, h, n, ratios, levels = G.make_inputs.BSTAGE[case]
    s, m = ratios[0]
    out, mp, d = O.morph_images(inp["c1"], inp["c2"], inp["gabor2"], inp["pts1"], inp["pts2"], s, m, levels, debug=True)
    for name in ("triMap", "mapx1", "mapy2", "trImg1", "trImg2", "lbmask", "lapBlend", "unsharp"):
        G.check(case, "f0_" + name, d[name])
    G.check(case, "f0_frame", out)


def test_pyramid_primitives():
    case = "b_256x256"
    l = G.full(case, "f1_l")
    if l is None:
        inp = G.bstage_inputs(case)
        pytest.skip("f1_l not stored in full")
    d0 = O.pyr_down(l)
    G.check(case, "f1_pyrDown0", d0)
    G.check(case, "f1_pyrUp0", O.pyr_up(d0, l.shape[1], l.shape[0]))
    d1 = O.pyr_down(d0)
    G.check(case, "f1_pyrDown1", d1)
    G.check(case, "f1_pyrUp1", O.pyr_up(d1, d0.shape[1], d0.shape[0]))
    G.check(case, "f1_maskDown0", O.pyr_down(G.full(case, "f1_lbmask")))


def test_unsharp_primitives():
    case = "b_64x48"
    lap = G.full(case, "f3_lapBlend")
    out, blur, med = O.unsharp(lap, 1.0 - np.sin(0.5 * np.pi), 0.3)
    G.check(case, "f3_usBlur", blur)
    G.check(case, "f3_usMedian", med)
    G.check(case, "f3_unsharp", out)


def test_prims_known_answers():
    inp = G.prims_inputs()
    w, h = int(inp["subdiv_rect"][0]), int(inp["subdiv_rect"][1])
    G.check("p_prims", "subdiv_tris", O.delaunay(w, h, inp["subdiv_pts"]))
    G.check("p_prims", "polys_map", O.paint_triangles(w, h, inp["polys"]))
    G.check("p_prims", "remap_dst", O.remap(inp["remap_src"], inp["remap_mx"], inp["remap_my"]))
    G.check("p_prims", "mats33_inv", O.invert33(inp["mats33"]))


def test_subdiv2d_opencv_known_answer():
    """Known-answer data of OpenCV's own test (OCV/imgproc/test/test_subdivision2d.cpp:9-58,
    regression_5788): these 65 landmark points in a 1500x2000 rect must give exactly 105 triangles,
    every vertex inside the rect."""
    pts = np.array([
        390, 802, 397, 883, 414, 963, 439, 1042, 472, 1113, 521, 1181, 591, 1238, 678, 1284, 771, 1292, 853, 1281,
        921, 1243, 982, 1191, 1030, 1121, 1059, 1038, 1072, 945, 1081, 849, 1082, 749, 459, 734, 502, 704, 554, 696,
        609, 698, 660, 707, 818, 688, 874, 661, 929, 646, 982, 653, 1026, 682, 740, 771, 748, 834, 756, 897,
        762, 960, 700, 998, 733, 1006, 766, 1011, 797, 999, 825, 987, 528, 796, 566, 766, 617, 763, 659, 794,
        619, 808, 569, 812, 834, 777, 870, 735, 918, 729, 958, 750, 929, 773, 882, 780, 652, 1102, 701, 1079,
        743, 1063, 774, 1068, 807, 1057, 852, 1065, 896, 1077, 860, 1117, 820, 1135, 783, 1141, 751, 1140, 706, 1130,
        675, 1102, 743, 1094, 774, 1094, 809, 1088, 878, 1082], np.float32).reshape(65, 2)
    tris = O.delaunay(1500, 2000, pts)
    assert len(tris) == 105
    t = tris.reshape(-1, 3, 2)
    assert (t[..., 0] >= 0).all() and (t[..., 0] < 1500).all() and (t[..., 1] >= 0).all() and (t[..., 1] < 2000).all()
