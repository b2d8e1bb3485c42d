"""The oracle on whole calls of the real poppy::morph (round-2 fixtures, captured from the compiled reference):
  x_dissolve_200x150   the no-match fallback expression img2*phase + img1*(1-phase) (src/poppy.hpp:129), 7 phases incl. -1
  a_256x256_phase01    phase == 0 / == 1 short-circuits (src/poppy.hpp:54-70)
  a_256x256_phase      one phase-mode frame from the raw pair
  a_512x512_chain30    BASELINE.json configs[0]: 512x512, 30 chained frames, from the raw pair (frames pinned by sha256)
  b_*_lv4 / _lv1       shallow pyramids (--pyramid 4 / 1)
and the host-side pieces of the library that carry the same logic (scheduler, printed morph distance).
"""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


def test_dissolve_expression_exact():
    case = "x_dissolve_200x150"
    inp = G.make_inputs.dissolve_inputs(case)
    for k, ph in enumerate(inp["phases"]):
        G.check(case, f"blend{k}", O.dissolve(inp["img1"], inp["img2"], float(ph)), what=f"phase {ph}")


def test_phase_zero_and_one_short_circuit():
    case = "a_256x256_phase01"
    inp = G.astage_inputs(case)
    n = int(inp["cfg"][0])
    frames = O.morph(inp["img1"], inp["img2"], n, phase=0.0)
    assert len(frames) == n == 3
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
        assert np.array_equal(f, inp["img1"])
    G.check(case, "phase0_frame", O.morph(inp["img1"], inp["img2"], 1, phase=1.0)[0])
    assert np.array_equal(G.full(case, "phase0_frame"), inp["img2"])


def test_scheduler_matches_library():
    from poppy_amd import capi
    L = capi.lib()
    for n in (1, 2, 30, 60, 480):
        for phase in (-1.0, 0.0, 0.25, 0.999, 1.0, 1.5):
            for j in sorted({0, 1, n // 2, n - 1}):
                assert O.frame_ratio(j, n, phase) == L.poppy_frame_ratio(j, n, phase), (j, n, phase)


@pytest.fixture(scope="module")
def setup_256():
    inp = G.astage_inputs("a_256x256_phase")
    return inp, O.pair_setup(inp["img1"], inp["img2"])


def test_pair_setup_from_raw_pair_256(setup_256):
    """Whole once-per-pair stage in the oracle (foreground, dft_detail2, ORB input, ORB, matcher, gabor2) from the raw pair."""
    inp, s = setup_256
    case = "a_256x256_chain"                      # same pair as a_256x256_phase; this case holds the intermediates
    ref = G.full(case, "detail")
    assert (s["detail"][0], s["detail"][1], s["nfeatures"]) == (ref[0], ref[1], int(ref[3]))
    G.check(case, "g1", s["g1"]); G.check(case, "g2", s["g2"])
    G.check(case, "kp1", s["kp1"]); G.check(case, "kp2", s["kp2"])
    G.check(case, "prepared1", s["points1"]); G.check(case, "prepared2", s["points2"])
    G.check(case, "gabor2", s["gabor2"])


def test_phase_mode_frame_from_raw_pair(setup_256):
    inp, s = setup_256
    frames = O.morph(inp["img1"], inp["img2"], int(inp["cfg"][0]), phase=float(inp["cfg"][1]), setup=s)
    assert len(frames) == 1
    G.check("a_256x256_phase", "frame0", frames[0])


def test_cfg1_512x512_30_chained_frames_from_raw_pair():
    """BASELINE.json configs[0] against the real poppy::morph: every one of the 30 frames by sha256."""
    case = "a_512x512_chain30"
    inp = G.astage_inputs(case)
    s = O.pair_setup(inp["img1"], inp["img2"])
    ref = G.full(case, "detail")
    assert s["nfeatures"] == int(ref[3]) and s["detail"] == (ref[0], ref[1])
    G.check(case, "g1", s["g1"])
    G.check(case, "prepared1", s["points1"]); G.check(case, "prepared2", s["points2"])
    G.check(case, "gabor2", s["gabor2"])
    assert s["distance"] == float(G.full(case, "printedMorphDist")[0])
    frames = O.morph(inp["img1"], inp["img2"], 30, setup=s)
    assert len(frames) == 30
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)


@pytest.mark.parametrize("case", ["b_640x480_lv4", "b_320x200_lv1"])
def test_shallow_pyramids(case):
    inp = G.bstage_inputs(case)
    w, h, n, ratios, levels = G.make_inputs.BSTAGE[case]
    for k, (sr, mr) in enumerate(ratios):
        out, mp, d = O.morph_images(inp["c1"], inp["c2"], inp["gabor2"], inp["pts1"], inp["pts2"], sr, mr, levels, debug=True)
        for name in ("lbmask", "lapBlend", "unsharp"):
            G.check(case, f"f{k}_{name}", d[name])
        G.check(case, f"f{k}_frame", out)


@pytest.mark.parametrize("case", ["a_512x512_chain30", "a_1920x1080_chain60", "a_3840x2160_phase"])
def test_printed_morph_distance_host(case):
    """poppy_printed_morph_distance (host half of --distance, src/poppy.hpp:142-159) on the reference's prepared point lists."""
    from poppy_amd import capi
    h, w = G.make_inputs.ASTAGE[case][1], G.make_inputs.ASTAGE[case][0]
    p1, p2 = G.full(case, "prepared1"), G.full(case, "prepared2")
    want = float(G.full(case, "printedMorphDist")[0])
    assert capi.printed_morph_distance(p1, p2, w, h) == want
    u1 = O.make_uniq(O.clip_points(p1, w, h)); u2 = O.make_uniq(O.clip_points(p2, w, h))
    k = min(len(u1), len(u2))
    assert O.morph_distance(u1[:k], u2[:k], w, h) == want


@pytest.mark.parametrize("case", ["a_320x180_photo", "a_256x192_textured"])
def test_whole_morph_on_non_synthetic_content(case):
    """The oracle on content that is not flat shapes (round 4): the reference's own sample photographs at 320 x 180 and hash-noise textures, against runs of the real
    reference on the same pixels: nfeatures, details, prepared point pairs, every chained frame bit for bit; gabor2 bit for bit outside the reference's DFT noise around exact
    zeros (flat regions of the photograph: |value| <= 1e-13 there, 0 in the oracle's direct sums)."""
    inp = G.astage_inputs(case)
    n = int(inp["cfg"][0])
    s = O.pair_setup(inp["img1"], inp["img2"])
    ref = G.full(case, "detail")
    assert s["nfeatures"] == int(ref[3]) and s["detail"] == (ref[0], ref[1])
    G.check(case, "prepared1", s["points1"])
    G.check(case, "prepared2", s["points2"])
    g = G.full(case, "gabor2")
    bad = g.view(np.uint32) != s["gabor2"].view(np.uint32)
    assert ((np.abs(g[bad]) <= 1e-12) & (np.abs(s["gabor2"][bad]) <= 1e-12)).all(), int(bad.sum())
    for j, f in enumerate(O.morph(inp["img1"], inp["img2"], n, setup=s)):
        G.check(case, f"frame{j}", f)
