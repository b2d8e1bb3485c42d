"""The oracle's once-per-pair core against the reference fixtures: ORB detect,
greedy point matcher, morph distance, threshold filter.  All comparisons are bit-exact, order included."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


def _bits_equal(a, b):
    return a.shape == b.shape and (np.ascontiguousarray(a).view(np.uint32) == np.ascontiguousarray(b).view(np.uint32)).all()


@pytest.mark.parametrize("case", ["o_256x256", "o_640x480"])
def test_orb_detect_bit_exact(case):
    inp = G.orb_inputs(case)
    w, h, nfs = G.make_inputs.ORB[case]
    for nf in nfs:
        for im in ("1", "2"):
            G.check(case, f"n{nf}_kp{im}", O.orb_detect(inp["g" + im], nf))


def test_fast_level0_matches_cv_fast():
    case = "o_256x256"
    inp = G.orb_inputs(case)
    _, fast = O.orb_detect(inp["g1"], 300, with_fast=True)
    ref = G.full(case, "fast_g1")               # cv::FAST(g1, 20, nonmax): x, y, size, angle, response, octave, class_id
    assert _bits_equal(fast, ref[:, [0, 1, 4]])
    assert (ref[:, 2] == 7).all() and (ref[:, 3] == -1).all()


@pytest.mark.slow
def test_orb_detect_1080p():
    case = "o_1920x1080"
    inp = G.orb_inputs(case)
    G.check(case, "n516_kp1", O.orb_detect(inp["g1"], 516))
    G.check(case, "n516_kp2", O.orb_detect(inp["g2"], 516))


@pytest.mark.parametrize("case", ["m_640x480", "m_1920x1080", "m_tol2"])
def test_point_matcher_bit_exact(case):
    inp = G.match_inputs(case)
    w, h, tol = int(inp["cfg"][0]), int(inp["cfg"][1]), float(inp["cfg"][2])
    G.check(case, "distanceMap", O.distance_map(inp["pts1"], inp["pts2"]))
    f1, f2 = O.filter_invalid(inp["pts1"], inp["pts2"], w, h)
    G.check(case, "filtered1", f1)
    G.check(case, "filtered2", f2)
    md = O.morph_distance(f1, f2, w, h)
    G.check(case, "initialMorphDist", np.array([md]))
    a, b = O.match_prepare(f1, f2, w, h, tol, md)
    G.check(case, "prepared1", a)
    G.check(case, "prepared2", b)
    # morph()'s printed distance after clip/uniq (src/poppy.hpp:142-159)
    u1 = O.make_uniq(O.clip_points(a, w, h)); u2 = O.make_uniq(O.clip_points(b, w, h))
    n = min(len(u1), len(u2))
    G.check(case, "finalMorphDist", np.array([O.morph_distance(u1[:n], u2[:n], w, h)]))


def test_real_pipeline_points_feed_the_matcher():
    """Keypoints of the real A stage (a_512x384_chain fixture) -> found/prepared point sets."""
    case = "a_512x384_chain"
    kp1, kp2 = G.full(case, "kp1"), G.full(case, "kp2")
    n = min(len(kp1), len(kp2))
    p1, p2 = kp1[:n, :2].copy(), kp2[:n, :2].copy()
    f1, f2 = O.filter_invalid(p1, p2, 512, 384)
    G.check(case, "found1", f1)
    G.check(case, "found2", f2)
    md = O.morph_distance(f1, f2, 512, 384)
    G.check(case, "initialMorphDist", np.array([md]))
    a, b = O.match_prepare(f1, f2, 512, 384, 1.0, md)
    G.check(case, "prepared1", a)
    G.check(case, "prepared2", b)


def test_real_pipeline_orb_on_reference_inputs():
    """ORB on the reference's own ORB input images (g1/g2 of the real pre-filter chain)."""
    case = "a_256x256_chain"
    nf = int(G.full(case, "detail")[3])
    G.check(case, "kp1", O.orb_detect(G.full(case, "g1"), nf))
    G.check(case, "kp2", O.orb_detect(G.full(case, "g2"), nf))


@pytest.mark.parametrize("case,nf", [("o_256x256", 300), ("o_640x480", 500)])
def test_orb_describe_and_bfmatch_bit_exact(case, nf):
    inp = G.orb_inputs(case)
    d = {}
    for im in ("1", "2"):
        d[im] = O.orb_describe(inp["g" + im], G.full(case, f"n{nf}_kp{im}"))
        G.check(case, f"n{nf}_desc{im}", d[im])
    G.check(case, f"n{nf}_bfmatch", O.hamming_match(d["1"], d["2"]))
    # descriptor-matching sketch (src/experiments.hpp:14-144): BFMatcher::knnMatch k=2 both ways, ratioTest, symmetryTest
    k12, k21 = O.hamming_knn2(d["1"], d["2"]), O.hamming_knn2(d["2"], d["1"])
    G.check(case, f"n{nf}_knn12", k12)
    G.check(case, f"n{nf}_knn21", k21)
    keep12, keep21, sym = O.ratio_symmetry(k12, k21, 0.7)
    G.check(case, f"n{nf}_ratio12", keep12)
    G.check(case, f"n{nf}_ratio21", keep21)
    G.check(case, f"n{nf}_sym", sym)


def test_knn2_edge_cases():
    """Ties keep the lower train index first; fewer than two train descriptors leave no second neighbour (K = min(2, n),
    OCV/core/src/batch_distance.cpp:286) and the ratio test then drops the query (experiments.hpp:30-33);
    0/0 is not > 0.7, so two exact duplicates survive the ratio test."""
    import numpy as np
    q = np.zeros((2, 32), np.uint8); q[1, 0] = 0xFF
    t = np.zeros((4, 32), np.uint8); t[0, 0] = 1; t[1, 1] = 1; t[2, 2] = 3; t[3, 0] = 0xFF
    k = O.hamming_knn2(q, t)
    assert k.tolist() == [[0, 1, 1, 1], [3, 0, 0, 7]]
    assert O.hamming_knn2(q, t[:1]).tolist() == [[0, 1, -1, -1], [0, 7, -1, -1]]
    assert O.hamming_knn2(q, t[:0]).tolist() == [[-1, -1, -1, -1]] * 2
    dup = np.zeros((2, 32), np.uint8)
    k12 = O.hamming_knn2(dup[:1], dup)
    assert k12.tolist() == [[0, 0, 1, 0]]
    keep12, keep21, sym = O.ratio_symmetry(k12, O.hamming_knn2(dup, dup[:1]))
    assert keep12.tolist() == [1] and keep21.tolist() == [0, 0] and len(sym) == 0
