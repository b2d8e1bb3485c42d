"""oracle/align.cpp against the reference's auto-align (src/matcher.cpp:133-244, src/transformer.cpp, src/procrustes.cpp) and the
OpenCV routines underneath it, run in this container through oracle/golden_gen (fixtures tests/golden/l_*)."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

CASES = ["l_317x211", "l_320x240", "l_640x480"]


def _inp(case):
    return G.make_inputs.align_inputs(case)


@pytest.mark.parametrize("case", CASES)
def test_warp_affine_and_rotation_matrix(case):
    inp = _inp(case)
    img = inp["img2"]
    h, w = img.shape[:2]
    G.check(case, "wa_t1", O.warp_affine(img, [1, 0, 5, 0, 1, -3]))
    G.check(case, "wa_t2", O.warp_affine(img, [1, 0, -40, 0, 1, 17]))
    c1 = (np.float32(w) / np.float32(3), np.float32(h) / np.float32(2))
    c2 = (np.float32(w) * np.float32(0.61), np.float32(h) * np.float32(0.27))
    rm1 = O.rotation_matrix(c1[0], c1[1], 12.0 + 1.0 / 3.0); rm2 = O.rotation_matrix(c2[0], c2[1], -100.0 / 3.0)
    G.check(case, "rm1", rm1)
    G.check(case, "rm2", rm2)
    G.check(case, "wa_r1", O.warp_affine(img, rm1))
    G.check(case, "wa_r2", O.warp_affine(img, rm2))
    G.check(case, "wa_a1", O.warp_affine(img, inp["aff"]))


@pytest.mark.parametrize("case", CASES)
def test_opencv_primitives_of_procrustes(case):
    inp = _inp(case)
    r = O.align_prims(inp["pts1"], inp["pts2"], G.full(case, "prim_svd_in"), G.full(case, "prim_svd_vt"))
    G.check(case, "prim_mean", r["mean"])
    G.check(case, "prim_sumsq", r["sumsq"])
    G.check(case, "prim_gemm", r["gemm"])
    G.check(case, "prim_svd_s", r["svd_w"])
    G.check(case, "prim_svd_u", r["svd_u"])
    G.check(case, "prim_svd_vt", r["svd_vt"])
    G.check(case, "prim_transform", r["transform"])
    G.check(case, "prim_persp", r["persp"])
    G.check(case, "prim_persp_pts", r["persp_pts"])


@pytest.mark.parametrize("case", CASES)
def test_procrustes(case):
    inp = _inp(case)
    r = O.procrustes(inp["pts1"], inp["pts2"])
    G.check(case, "pc_rotation", r["rotation"])
    sc = G.full(case, "pc_scalars")
    assert np.float32(sc[0]) == r["scale"] and np.float32(sc[1]) == r["error"]
    G.check(case, "pc_yprime", r["yprime"])
    G.check(case, "pc_translation", r["translation"])


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("step,pf", [("retranslate", "rt"), ("reprocrustes", "rp"), ("rerotate", "rr")])
def test_transformer_steps(case, step, pf):
    inp = _inp(case)
    h, w = inp["img2"].shape[:2]
    assert O.morph_distance(inp["pts1"], inp["pts2"], w, h) == G.full(case, "md0")[0]
    img, p2, d = O.align_step(step, inp["img2"], inp["pts1"], inp["pts2"])
    G.check(case, pf + "_pts2", p2)
    assert d == G.full(case, pf + "_dist")[0]
    G.check(case, pf + "_img", img)


@pytest.mark.parametrize("case", CASES)
def test_auto_align(case):
    inp = _inp(case)
    img, p2, d = O.align_step("auto", inp["img2"], inp["pts1"], inp["pts2"])
    G.check(case, "aa_pts1", inp["pts1"])
    G.check(case, "aa_pts2", p2)
    assert d == G.full(case, "aa_dist")[0]
    G.check(case, "aa_img", img)
