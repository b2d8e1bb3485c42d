"""GPU parity of the rest of the pre-ORB chain (src/extractor.cpp:33-83, src/poppy.hpp:119-122) through the C ABI,
against the real reference's outputs (tests/golden/a_*, d_*).

Everything is compared bit for bit: unsharp sigma 2 + grey; dft_detail2 (the RMS of raw float bytes of the spectrum,
which only cv::dft's exact factorisation and butterfly order reproduces); the ORB input images; gabor2; the prepared
point pairs and the chained frames of the real poppy::morph.  One intermediate is allowed a tolerance: the Gabor mean
BEFORE it is multiplied and quantised.  The reference computes each Gabor plane through double-precision DFTs and
rounds it to float once; the kernel accumulates the same sum directly in double and rounds once.  Where the exact
value is ~0 the DFT's own noise (1e-13) shows up as a different tiny float; those values vanish in the next stage."""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

CASES = ["a_256x256_chain", "a_512x384_chain"]


@pytest.fixture(scope="module")
def ctx():
    from poppy_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", CASES)
def test_orb_input_chain_from_reference_good_features(ctx, case):
    gf1 = G.full(case, "goodFeatures1")
    r = ctx.orb_input(gf1)
    G.check(case, "us1", r["us"])                                   # unsharp_mask(sigma 2) + BGR2GRAY: exact
    gb = G.full(case, "gb1")
    assert np.abs(r["gb"] - gb).max() <= 1e-12                      # Gabor mean: once-rounded exact sums vs double-DFT noise (measured 1.3e-13)
    G.check(case, "g1", r["g"])                                     # the ORB input image: exact
    assert r["detail"] == G.full(case, "detail")[0]                 # dft_detail2: exact (cv::dft's operation order is reproduced)


def test_gabor_field_vs_reference(ctx):
    case = "a_256x256_chain"
    inp = G.astage_inputs(case)
    gab = ctx.gabor_field(inp["img2"])
    G.check(case, "gabor2", gab)                                    # exact


@pytest.mark.parametrize("case", CASES)
def test_pair_begin_from_raw_images_reproduces_poppy_morph(case):
    """Whole pipeline on the GPU from the raw pair: nfeatures, prepared point pairs and every chained frame of the real
    poppy::morph, bit for bit."""
    from poppy_amd import capi
    inp = G.astage_inputs(case)
    c = capi.Context(0, number_of_frames=int(inp["cfg"][0]))
    nf, det = c.pair_begin(inp["img1"], inp["img2"])
    ref = G.full(case, "detail")
    assert nf == int(ref[3]) and det == (ref[0], ref[1])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1)
    G.check(case, "prepared2", p2)
    frames = c.morph_frames(-1.0)
    assert len(frames) == int(inp["cfg"][0])
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
    c.close()


@pytest.mark.parametrize("case", CASES)
def test_pair_begin_with_the_chains_one_after_the_other(case):
    """poppy_hip_set_setup_chains(1) — what the contexts of a pool of three or more use: the two images' chains one after the other instead of side by side.  The same
    pair state as the default order and as the real poppy::morph: detail values, nfeatures, prepared point pairs, gabor2, the last frame."""
    from poppy_amd import capi
    inp = G.astage_inputs(case)
    c = capi.Context(0, number_of_frames=int(inp["cfg"][0]))
    nf0, det0 = c.pair_begin(inp["img1"], inp["img2"])
    a0, b0 = c.pair_points()
    g0 = c.fetch("gabor2")
    c.set_setup_chains(True)
    nf, det = c.pair_begin(inp["img1"], inp["img2"])
    ref = G.full(case, "detail")
    assert nf == nf0 == int(ref[3]) and det == det0 == (ref[0], ref[1])
    p1, p2 = c.pair_points()
    assert np.array_equal(p1, a0) and np.array_equal(p2, b0)
    G.check(case, "prepared1", p1)
    G.check(case, "prepared2", p2)
    assert np.array_equal(c.fetch("gabor2").view(np.uint32), g0.view(np.uint32))
    frames = c.morph_frames(-1.0)
    G.check(case, f"frame{len(frames) - 1}", frames[-1])
    c.close()


@pytest.mark.parametrize("case", ["a_256x256_radial", "a_320x200_radial"])
def test_pair_begin_with_radial_mask_reproduces_poppy_morph(case):
    """Settings::enable_radial_mask (the CLI's --radial; src/extractor.cpp:178-197, src/draw.cpp:21-38) against the REAL reference run with the
    option (fixtures of round 4): the mask table, both foregrounds, nfeatures, the prepared point pairs and every frame of poppy::morph, bit for bit."""
    from poppy_amd import capi
    inp = G.astage_inputs(case)
    assert int(inp["cfg"][3]) == 2                                     # flags: bit 1 = the radial mask
    h, w = inp["img1"].shape[:2]
    G.check(case, "radialMask", capi.radial_mask(w, h))
    c = capi.Context(0, number_of_frames=int(inp["cfg"][0]), enable_radial_mask=1)
    G.check(case, "goodFeatures1", c.foreground(inp["img1"]))
    G.check(case, "goodFeatures2", c.foreground(inp["img2"]))
    nf, det = c.pair_begin(inp["img1"], inp["img2"])
    ref = G.full(case, "detail")
    assert nf == int(ref[3]) and det == (ref[0], ref[1])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1)
    G.check(case, "prepared2", p2)
    frames = c.morph_frames(-1.0)
    assert len(frames) == int(inp["cfg"][0])
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
    c.close()
    plain = capi.Context(0, number_of_frames=int(inp["cfg"][0]))       # and the option changes the outcome: without it other features are found
    assert not np.array_equal(plain.foreground(inp["img1"]), G.full(case, "goodFeatures1"))
    plain.close()


def _gabor2_matches(ref, got):
    """gabor2 against the reference's: bit for bit, with two documented exceptions.  (1) Where the exact correlation sum is 0 (flat regions of a
    photograph) the reference holds the noise of its double-precision DFTs (|value| <= 1e-13) and a restatement holds 0 or its own noise — both vanish in
    m2 = 1 - gray(gabor2).  (2) Where the exact sum sits within the DFT noise of a float rounding boundary, a double-precision FFT correlation (the reference's, the library's
    default form) can land on either neighbour: one ulp, about one float in 1e8.  Returns (ok, floats of kind 1, floats of kind 2)."""
    bad = ref.view(np.uint32) != got.view(np.uint32)
    zero_noise = bad & (np.abs(ref) <= 1e-12) & (np.abs(got) <= 1e-12)
    rest = bad & ~zero_noise
    one_ulp = rest & (np.abs(ref.view(np.int32).astype(np.int64) - got.view(np.int32).astype(np.int64)) <= 1)
    return bool((rest == one_ulp).all()) and int(one_ulp.sum()) <= 4, int(zero_noise.sum()), int(one_ulp.sum())


@pytest.mark.parametrize("case", ["a_320x180_photo", "a_256x192_textured"])
def test_whole_morph_on_non_synthetic_content(case):
    """The whole of poppy::morph on content that is not flat shapes — the reference's own sample photographs (images/amir1.jpg / amir2.jpg, committed as
    pixels) at 320 x 180, hash-noise textures at 256 x 192 — against runs of the REAL reference on the same pixels (fixtures of round 4): nfeatures, both
    details, the prepared point pairs, every chained frame and a phase-mode frame bit for bit; gabor2 bit for bit outside the reference's own DFT noise
    around exact zeros.  (Timing the set-up on such content found the detector's candidate lists too short for noise-like images; the medians' whole-wave
    skips and the detector's candidate guess are content-dependent code paths.)"""
    from poppy_amd import capi
    inp = G.astage_inputs(case)
    n = int(inp["cfg"][0])
    c = capi.Context(0, number_of_frames=n)
    nf, det = c.pair_begin(inp["img1"], inp["img2"])
    ref = G.full(case, "detail")
    assert nf == int(ref[3]) and det == (ref[0], ref[1])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1)
    G.check(case, "prepared2", p2)
    gab = c.fetch("gabor2")
    ok, n_zero, n_ulp = _gabor2_matches(G.full(case, "gabor2"), gab)
    assert ok and n_ulp == 0, (n_zero, n_ulp)
    cd = capi.Context(0, number_of_frames=n)                           # the direct double sums: the FFT form must equal them in every bit
    cd.set_gabor_direct(True)
    cd.pair_begin(inp["img1"], inp["img2"])
    assert np.array_equal(gab.view(np.uint32), cd.fetch("gabor2").view(np.uint32))
    cd.close()
    frames = c.morph_frames(-1.0)
    assert len(frames) == n
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
    c.close()
    if len(inp["cfg"]) > 4:                                            # a phase-mode frame of the same pair (number_of_frames = 1)
        c1 = capi.Context(0, number_of_frames=1)
        rc, fr, _ = c1.morph(inp["img1"], inp["img2"], phase=float(inp["cfg"][4]))
        assert rc == 0 and len(fr) == 1
        G.check(case, "phase0_frame", fr[0])
        c1.close()


def test_pair_begin_descriptors_mode():
    """Opt-in descriptor matching: the point pairs must be what the reference's sketch (knnMatch k=2 both ways,
    ratioTest 0.7, symmetryTest; src/experiments.hpp:14-144) selects from the reference pipeline's own keypoints, in query
    order, plus the four corners; the frames then come from the ordinary per-frame path."""
    import oracle_lib as O
    from poppy_amd import capi
    case = "a_256x256_chain"
    inp = G.astage_inputs(case)
    c = capi.Context(0, number_of_frames=4)
    nf = c.pair_begin_descriptors(inp["img1"], inp["img2"], 0.7)
    assert nf == int(G.full(case, "detail")[3])
    p1, p2 = c.pair_points()
    kp1, kp2 = G.full(case, "kp1"), G.full(case, "kp2")              # cv::KeyPoint rows of the reference run
    d1 = O.orb_describe(G.full(case, "g1"), kp1); d2 = O.orb_describe(G.full(case, "g2"), kp2)
    _, _, sym = O.ratio_symmetry(O.hamming_knn2(d1, d2), O.hamming_knn2(d2, d1), 0.7)
    h, w = inp["img1"].shape[:2]
    corners = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]], np.float32)
    want1 = np.concatenate([kp1[sym[:, 0], :2], corners]); want2 = np.concatenate([kp2[sym[:, 1], :2], corners])
    assert len(sym) > 20
    assert np.array_equal(p1, want1.astype(np.float32)) and np.array_equal(p2, want2.astype(np.float32))
    frames = c.morph_frames(-1.0)                                      # the ordinary per-frame path on those pairs
    assert len(frames) == 4 and frames[0].shape == inp["img1"].shape
    assert np.array_equal(frames[-1], inp["img2"]) or np.abs(frames[-1].astype(int) - inp["img2"].astype(int)).mean() < 20
    c.close()


@pytest.mark.parametrize("w,h,seed", [(97, 61, 2), (64, 130, 3)])
def test_orb_input_ragged_vs_oracle(ctx, w, h, seed):
    import oracle_lib as O
    from poppy_amd import synth
    gf = synth.textured_gray(w, h, seed)
    r = ctx.orb_input(gf)
    us = O.orb_unsharp_gray(gf)
    assert np.array_equal(r["us"].view(np.uint32), us.view(np.uint32))           # exact stage
    gb = O.gabor_filter_direct(us, 31, O.gabor_bank(31, 5, 2))
    assert np.array_equal(r["gb"].view(np.uint32), gb.view(np.uint32))             # both: exact products, double sums, one rounding


@pytest.mark.parametrize("w,h,seed", [(20, 12, 3), (9, 40, 4), (35, 34, 8)])
def test_gabor_banks_fft_small_images(ctx, w, h, seed):
    """Images smaller than the kernel radius / the 34-pixel FFT block: the patch is reflected several times (reflect-101, as
    filter2D's borderInterpolate does); both forms of the banks must agree there too."""
    from poppy_amd import synth
    gf = synth.textured_gray(w, h, seed)
    bgr = synth.textured_bgr(w, h, seed + 100)
    try:
        ctx.set_gabor_direct(False)
        a, fa = ctx.orb_input(gf), ctx.gabor_field(bgr)
        ctx.set_gabor_direct(True)
        b, fb = ctx.orb_input(gf), ctx.gabor_field(bgr)
    finally:
        ctx.set_gabor_direct(False)
    assert np.array_equal(a["gb"].view(np.uint32), b["gb"].view(np.uint32)) and np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
    assert np.array_equal(a["g"], b["g"])


@pytest.mark.parametrize("w,h,seed", [(640, 480, 5), (1001, 333, 6), (1920, 1080, 7)])
def test_gabor_banks_fft_vs_direct_sums(ctx, w, h, seed):
    """The two forms of the Gabor banks (tiled double-precision FFTs, the default; direct double sums) give the same float planes in
    every bit: both evaluate the reference's double-precision correlation, ~1e-15 apart before the one rounding to float (the
    reference's own DFT noise is ~1e-13), and the FFT form re-forms as a direct sum every plane value that lies close enough to a float
    rounding boundary (or to zero) for that distance to matter (until round 4 such a value differed by one ulp: 1 in 1.3e8 over these
    three sizes, the 13 x 13 bank at 1920x1080)."""
    from poppy_amd import capi, synth
    gf = synth.textured_gray(w, h, seed)
    bgr = synth.textured_bgr(w, h, seed + 100)
    try:
        ctx.set_gabor_direct(False)
        capi.gabor_doubt()
        a, fa = ctx.orb_input(gf), ctx.gabor_field(bgr)
        near_zero, near_mid, redone = capi.gabor_doubt()
        ctx.set_gabor_direct(True)
        b, fb = ctx.orb_input(gf), ctx.gabor_field(bgr)
    finally:
        ctx.set_gabor_direct(False)
    assert 0 < redone <= near_zero + near_mid < 1e-3 * 16 * (a["gb"].size + fa.size), (near_zero, near_mid, redone)   # the hand-over to the direct sums ran, on few pixels
    n31 = int((a["gb"].view(np.uint32) != b["gb"].view(np.uint32)).sum())
    n13 = int((fa.view(np.uint32) != fb.view(np.uint32)).sum())
    d31, d13 = float(np.abs(a["gb"] - b["gb"]).max()), float(np.abs(fa - fb).max())
    assert n31 == 0 and n13 == 0, (n31, n13, d31, d13)
    assert np.array_equal(a["g"], b["g"])


@pytest.mark.parametrize("case", sorted(G.make_inputs.DETAIL))
def test_dft_detail2_exact(ctx, case):
    """dft_detail2 is the RMS of raw float bytes of the spectrum: only cv::dft's exact operation order reproduces it."""
    gray = G.make_inputs.detail_inputs(case)["gray"]
    got = ctx.orb_input(gray)["detail"]
    want = float(G.full(case, "detail")[0])
    assert got == want, f"{case}: {got!r} != {want!r} (rel {abs(got - want) / want:.2e})"


@pytest.mark.parametrize("case", sorted(G.make_inputs.MARGIN))
def test_blur_margin_exact(ctx, case):
    inp = G.make_inputs.margin_inputs(case)
    G.check(case, "padded", ctx.blur_margin(inp["img"], int(inp["cfg"][0]), int(inp["cfg"][1])))
