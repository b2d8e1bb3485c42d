"""GPU parity of the auto-align through the C ABI: cv::warpAffine kernel, the three Transformer steps,
Matcher::autoAlign, and poppy_hip_pair_begin with enable_auto_align — against the reference fixtures (tests/golden/l_*,
a_*_align) and the oracle.  Everything is compared bit for bit."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from poppy_amd import capi, synth

pytestmark = pytest.mark.gpu
CASES = ["l_317x211", "l_320x240", "l_640x480"]


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", CASES)
def test_warp_affine_vs_opencv(ctx, case):
    inp = G.make_inputs.align_inputs(case)
    img = inp["img2"]
    G.check(case, "wa_t1", ctx.warp_affine(img, [1, 0, 5, 0, 1, -3]))
    G.check(case, "wa_t2", ctx.warp_affine(img, [1, 0, -40, 0, 1, 17]))
    G.check(case, "wa_r1", ctx.warp_affine(img, G.full(case, "rm1")))
    G.check(case, "wa_r2", ctx.warp_affine(img, G.full(case, "rm2")))
    G.check(case, "wa_a1", ctx.warp_affine(img, inp["aff"]))


def test_warp_affine_odd_maps_vs_oracle(ctx):
    """Singular, mirrored, far-away and strongly scaled maps, ragged size."""
    img = synth.textured_bgr(203, 97, 8)
    for M in ([0, 0, 10, 0, 0, 20], [-1, 0, 202, 0, 1, 0], [1, 0, 1e7, 0, 1, -1e7], [3.7, 0.2, -100, -0.3, 0.25, 40],
              [1, 0, 0.5, 0, 1, 0.25], [1e-9, 0, 0, 0, 1e-9, 0]):
        assert np.array_equal(ctx.warp_affine(img, M), O.warp_affine(img, M)), M


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("step,pf", [("retranslate", "rt"), ("reprocrustes", "rp"), ("rerotate", "rr")])
def test_transformer_steps_vs_reference(ctx, case, step, pf):
    inp = G.make_inputs.align_inputs(case)
    img, p2, d = ctx.align(step, inp["img2"], inp["pts1"], inp["pts2"])
    G.check(case, pf + "_pts2", p2)
    assert d == G.full(case, pf + "_dist")[0]
    G.check(case, pf + "_img", img)


@pytest.mark.parametrize("case", CASES)
def test_auto_align_vs_reference(ctx, case):
    inp = G.make_inputs.align_inputs(case)
    img, p2, d = ctx.align("auto", inp["img2"], inp["pts1"], inp["pts2"])
    G.check(case, "aa_pts2", p2)
    assert d == G.full(case, "aa_dist")[0]
    G.check(case, "aa_img", img)


@pytest.mark.parametrize("serial", [False, True])
@pytest.mark.parametrize("case", ["a_256x256_align", "a_384x288_align"])
def test_pair_begin_with_auto_align_reproduces_poppy_morph(case, serial):
    """poppy::morph with Settings::enable_auto_align from the raw pair: prepared points and every frame, bit for bit — with the two images' chains side by side
    (a context's default) and one after the other (poppy_hip_set_setup_chains: what the contexts of a pool of three or more use)."""
    inp = G.astage_inputs(case)
    c = capi.Context(0, number_of_frames=int(inp["cfg"][0]), enable_auto_align=1)
    c.set_setup_chains(serial)
    nf, _ = c.pair_begin(inp["img1"], inp["img2"])
    assert nf == int(G.full(case, "detail")[3])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1)
    G.check(case, "prepared2", p2)
    hh, ww = inp["img2"].shape[:2]
    G.check(case, "corrected2", c.pair_corrected2(ww, hh))          # what poppy::morph returns to its caller
    frames = c.morph_frames(-1.0)
    assert len(frames) == int(inp["cfg"][0])
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
    c.close()


def test_align_entry_points_argument_errors_and_strides(ctx):
    """Error behaviour (status codes, no exit / throw) and row strides of the host-facing align entry points."""
    import ctypes as C
    L = capi.lib()
    img = synth.textured_bgr(64, 40, 3)
    pts = np.array([[5, 5], [50, 6], [8, 30]], np.float32)
    d = C.c_double(0)
    rc = L.poppy_hip_auto_align(ctx.h, img.ctypes.data, 64 * 3, 64, 40, pts.ctypes.data, pts.copy().ctypes.data, 3, C.byref(d))
    assert rc == -1 and b"4 point pairs" in L.poppy_hip_last_error(ctx.h)          # POPPY_E_ARG: fewer than 4 pairs
    assert L.poppy_hip_align_step(ctx.h, 7, img.ctypes.data, 64 * 3, 64, 40, pts.ctypes.data, pts.ctypes.data, 3, C.byref(d)) == -1
    assert L.poppy_hip_warp_affine(ctx.h, img.ctypes.data, 10, 64, 40, None, img.ctypes.data, 64 * 3) == -1
    # padded rows in, padded rows out
    M = np.array([0.9, 0.1, 3.0, -0.1, 0.95, 1.5], np.float64)
    src = np.zeros((40, 80 * 3), np.uint8); src[:, :64 * 3] = img.reshape(40, -1)
    dst = np.full((40, 72 * 3), 7, np.uint8)
    assert L.poppy_hip_warp_affine(ctx.h, src.ctypes.data, 80 * 3, 64, 40, M.ctypes.data, dst.ctypes.data, 72 * 3) == 0
    assert np.array_equal(dst[:, :64 * 3].reshape(40, 64, 3), O.warp_affine(img, M)) and (dst[:, 64 * 3:] == 7).all()
