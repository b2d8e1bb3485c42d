"""The oracle against the known-answer data that the REFERENCE'S OWN TESTS hold (vendored OpenCV 4.6.0 test sources; OCV =
third/opencv-4.6.0/modules) — the pins that do not depend on anything generated in this repository.  The literal values
below are inputs and expected outputs of those tests (data, not code).

  routine of the oracle            reference-held vector
  Subdiv2D order / count           OCV/imgproc/test/test_subdivision2d.cpp:9-58  (tests/test_oracle_bstage.py)
  fixed-point Gaussian taps        OCV/imgproc/test/test_smooth_bitexact.cpp:14-27 (vU8)
  fillConvexPoly raster            OCV/imgproc/test/test_drawing.cpp:432-462 (fillconvexpoly_clipping)
  BFMatcher(NORM_HAMMING)          OCV/features2d/test/test_matchers_algorithmic.cpp:604-619 (issue_11855)
  convexHull                       OCV/imgproc/test/test_convhull.cpp:2280-2308 (overflow: int hull == float hull)
  3x3 invert                       OCV/core/test/test_math.cpp:2727-2740 (Core_Invert.small)
  pyrDown                          OCV/imgproc/test/test_filter.cpp:2318-2324 (issue_12961: zeros stay zeros)
"""
import numpy as np

import oracle_lib as O


def test_fixed_point_gaussian_kernels():
    one = 256
    vu8 = [  # (size, sigma, taps) — test_smooth_bitexact.cpp:14-27
        (1, 0, [one]),
        (3, 0, [one >> 2, one >> 1, one >> 2]),
        (5, 0, [one >> 4, one >> 2, 6 * (one >> 4), one >> 2, one >> 4]),
        (7, 0, [one >> 5, 7 * (one >> 6), 7 * (one >> 5), 9 * (one >> 5), 7 * (one >> 5), 7 * (one >> 6), one >> 5]),
        (9, 0, [4, 13, 30, 51, 60, 51, 30, 13, 4]),
        (3, 1.75, [81, 94, 81]),
        (3, 0.875, [65, 126, 65]),
        (5, 0.375, [0, 7, 242, 7, 0]),
        (5, 0.75, [4, 56, 136, 56, 4]),
    ]
    for n, sigma, want in vu8:
        assert O.gaussian_taps_fx(n, sigma).tolist() == want, (n, sigma)
    # the two kernels the hot path uses sum to one in fixed point like every kernel of that generator
    for n, sigma in ((23, 1.0), (127, 6.0)):
        assert int(O.gaussian_taps_fx(n, sigma).sum()) == 256


def test_fill_convex_poly_clipping():
    # a 10x10 image, the polygon (1,1) (5,1) (5,8) (1,8): 40 pixels are painted
    img = np.zeros((10, 10), np.int32)
    out = O.fill_convex(img, [(1, 1), (5, 1), (5, 8), (1, 8)], 255)
    assert int(np.count_nonzero(out)) == 40
    assert (out[1:9, 1:6] == 255).all()
    # second half of that test: a polygon reaching far outside the image must not write outside (it crashed once)
    big = O.fill_convex(np.zeros((10, 10), np.int32), [(2, 2), (10, 2), (10, 16), (2, 16)], 7)
    assert (big[2:, 2:] == 7).all() and int(np.count_nonzero(big)) == 64


def test_bfmatcher_hamming_crosscheck_known_answer():
    # sources / targets of issue_11855, zero-padded to the oracle's 32-byte descriptors (padding adds no distance)
    src = np.zeros((2, 32), np.uint8); src[0, :3] = [1, 1, 0]; src[1, :3] = [1, 1, 1]
    tgt = np.zeros((2, 32), np.uint8); tgt[0, :3] = [1, 1, 1]; tgt[1, :3] = [0, 0, 0]
    fwd = O.hamming_match(src, tgt)            # rows queryIdx, trainIdx, distance
    bwd = O.hamming_match(tgt, src)
    # BFMatcher(NORM_HAMMING, crossCheck = true).knnMatch(k = 1, compactResult): a pair survives when each side is the
    # other's nearest neighbour
    kept = [(int(q), int(t), int(d)) for q, t, d in fwd if int(bwd[t][1]) == int(q)]
    assert kept == [(1, 0, 0)]
    assert fwd.tolist() == [[0, 0, 1], [1, 0, 0]]


def test_convex_hull_of_large_coordinates():
    pts = np.array([[14763, 2890], [14388, 72088], [62810, 72274], [63166, 3945], [56782, 3945], [56763, 3077],
                    [34666, 2965], [34547, 2953], [34508, 2866], [34429, 2965]], np.float32)
    hull = O.convex_hull(pts)

    def cross(o, a, b):                         # exact: Python integers
        return (int(a[0]) - int(o[0])) * (int(b[1]) - int(o[1])) - (int(a[1]) - int(o[1])) * (int(b[0]) - int(o[0]))
    # the exact hull by Andrew's monotone chain over the integer coordinates
    p = sorted(set((int(x), int(y)) for x, y in pts))
    lower, upper = [], []
    for q in p:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], q) <= 0:
            lower.pop()
        lower.append(q)
    for q in reversed(p):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], q) <= 0:
            upper.pop()
        upper.append(q)
    want = set(lower[:-1] + upper[:-1])
    got = [(int(x), int(y)) for x, y in hull]
    assert set(got) == want and len(got) == len(want)
    # one orientation all the way round (the int and the float overloads of cv::convexHull return the same cycle)
    turns = [cross(got[i - 2], got[i - 1], got[i]) for i in range(len(got))]
    assert all(t > 0 for t in turns) or all(t < 0 for t in turns)


def test_invert_3x3_small_matrix():
    a = np.array([[2.42104644730331, 1.81444796521479, -3.98072565304758],
                  [0, 7.08389214348967e-3, 5.55326770986007e-3],
                  [0, 0, 7.44556154284261e-3]], np.float32)
    b = (a.T @ a).astype(np.float32)
    c = O.invert33(b[None])[0]
    assert np.abs(b.astype(np.float64) @ c.astype(np.float64) - np.eye(3)).max() < 0.1


def test_pyrdown_of_zeros_is_zero():
    assert not O.pyr_down(np.zeros((9, 9), np.float32)).any()
