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
  cvRound (double and float)       OCV/core/test/test_arithm.cpp:1679-1690 (Core_round.CvRound: halves to even)
  large-kernel filter2D            OCV/imgproc/test/test_filter.cpp:2147-2221 (dftFilter2d_regression_10683: literal 24 x 24 in / out, tol 2)
                                   OCV/imgproc/test/test_filter.cpp:2223-2282 (dftFilter2d_regression_13179: a 13 x 13 getGaborKernel on a 16 x 16 ROI, literal in / out, tol 2)
  float GaussianBlur               OCV/imgproc/test/test_filter.cpp:2358-2367 (regression_11303: a blurred constant image stays the constant, 2115 x 211 — as a property
                                   of the oracle's own two float Gaussians; the reference call's sigma 8.64 kernel is not on the path)
  medianBlur                       OCV/imgproc/test/test_filter.cpp:2284-2294 (hires_regression_13409: the median of a region = the region of the median, images over 1024)
Not usable: Multiply.FloatingPointRounding (test_arithm.cpp:1649-1657) pins cv::multiply(Mat, Scalar) in double; the path has no such call
(its scaled accumulate is MatExpr `flow * s`, a convertTo with a float scale: oracle/prefilter.cpp accumulate_scaled_u8).
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


def test_cv_round_known_answers():
    """Core_round.CvRound (OCV/core/test/test_arithm.cpp:1679-1690): cvRound rounds halves to even.  Every float -> integer step of the path
    (remap's cvRound(x * 32), convertTo(CV_8U), the 8-bit scaled accumulate) goes through the oracle's cv_round / cv_round_f."""
    import ctypes as C
    L = O.lib()
    L.orc_round_d.restype = C.c_int; L.orc_round_d.argtypes = [C.c_double]
    L.orc_round_f.restype = C.c_int; L.orc_round_f.argtypes = [C.c_float]
    for x, want in ((2.0, 2), (2.1, 2), (-2.1, -2), (2.8, 3), (-2.8, -3), (2.5, 2), (3.5, 4), (-2.5, -2), (-3.5, -4)):
        assert L.orc_round_d(x) == want, x
        assert L.orc_round_f(x) == want, x          # all nine values are exact in float: the float overload agrees


_SRC_10683 = np.array([
    0, 40, 0, 0, 255, 0, 0, 78, 131, 0, 196, 0, 255, 0, 0, 0, 0, 255, 70, 0, 255, 0, 0, 0,
    0, 0, 255, 204, 0, 0, 255, 93, 255, 0, 0, 255, 12, 0, 0, 0, 255, 121, 0, 255, 0, 0, 0, 255,
    0, 178, 0, 25, 67, 0, 165, 0, 255, 0, 0, 181, 151, 175, 0, 0, 32, 0, 0, 255, 165, 93, 0, 255,
    255, 255, 0, 0, 255, 126, 0, 0, 0, 0, 133, 29, 9, 0, 220, 255, 0, 142, 255, 255, 255, 0, 255, 0,
    255, 32, 255, 0, 13, 237, 0, 0, 0, 0, 0, 19, 90, 0, 0, 85, 122, 62, 95, 29, 255, 20, 0, 0,
    0, 0, 166, 41, 0, 48, 70, 0, 68, 0, 255, 0, 139, 7, 63, 144, 0, 204, 0, 0, 0, 98, 114, 255,
    105, 0, 0, 0, 0, 255, 91, 0, 73, 0, 255, 0, 0, 0, 255, 198, 21, 0, 0, 0, 255, 43, 153, 128,
    0, 98, 26, 0, 101, 0, 0, 0, 255, 0, 0, 0, 255, 77, 56, 0, 241, 0, 169, 132, 0, 255, 186, 255,
    255, 87, 0, 1, 0, 0, 10, 39, 120, 0, 23, 69, 207, 0, 0, 0, 0, 84, 0, 0, 0, 0, 255, 0,
    255, 0, 0, 136, 255, 77, 247, 0, 67, 0, 15, 255, 0, 143, 0, 243, 255, 0, 0, 238, 255, 0, 255, 8,
    42, 0, 0, 255, 29, 0, 0, 0, 255, 255, 255, 75, 0, 0, 0, 255, 0, 0, 255, 38, 197, 0, 255, 87,
    0, 123, 17, 0, 234, 0, 0, 149, 0, 0, 255, 16, 0, 0, 0, 255, 0, 255, 0, 38, 0, 114, 255, 76,
    0, 0, 8, 0, 255, 0, 0, 0, 220, 0, 11, 255, 0, 0, 55, 98, 0, 0, 0, 255, 0, 175, 255, 110,
    235, 0, 175, 0, 255, 227, 38, 206, 0, 0, 255, 246, 0, 0, 123, 183, 255, 0, 0, 255, 0, 156, 0, 54,
    0, 255, 0, 202, 0, 0, 0, 0, 157, 0, 255, 63, 0, 0, 0, 0, 0, 255, 132, 0, 255, 0, 0, 0,
    0, 0, 0, 255, 0, 0, 128, 126, 0, 243, 46, 7, 0, 211, 108, 166, 0, 0, 162, 227, 0, 204, 0, 51,
    255, 216, 0, 0, 43, 0, 255, 40, 188, 188, 255, 0, 0, 255, 34, 0, 0, 168, 0, 0, 0, 35, 0, 0,
    0, 80, 131, 255, 0, 255, 10, 0, 0, 0, 180, 255, 209, 255, 173, 34, 0, 66, 0, 49, 0, 255, 83, 0,
    0, 204, 0, 91, 0, 0, 0, 205, 84, 0, 0, 0, 92, 255, 91, 0, 126, 0, 185, 145, 0, 0, 9, 0,
    255, 0, 0, 255, 255, 0, 0, 255, 0, 0, 216, 0, 187, 221, 0, 0, 141, 0, 0, 209, 0, 0, 255, 0,
    255, 0, 0, 154, 150, 0, 0, 0, 148, 0, 201, 255, 0, 255, 16, 0, 0, 160, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 255, 0, 255, 0, 255, 0, 255, 198, 255, 147, 131, 0, 255, 202, 0, 0, 0, 0, 255, 0,
    0, 0, 0, 164, 181, 0, 0, 0, 69, 255, 31, 0, 255, 195, 0, 0, 255, 164, 109, 0, 0, 202, 0, 206,
    0, 0, 61, 235, 33, 255, 77, 0, 0, 0, 0, 85, 0, 228, 0, 0, 0, 0, 255, 0, 0, 5, 255, 255], np.float32).reshape(24, 24)      # the literal 24 x 24 image of regressions 10683 and 13179


def test_dft_filter2d_regression_10683():
    """Imgproc_Filter2D.dftFilter2d_regression_10683 (OCV/imgproc/test/test_filter.cpp:2147-2221): a 24 x 24 8-bit image filtered with a
    12 x 12 box kernel through filter2D's DFT path, literal input and expected output, tolerance 2.  filter2D with a kernel this large is
    the correlation the Gabor banks take (OCV/imgproc/src/filter.dispatch.cpp:1291-1292 -> crossCorr); the oracle's form of it is
    gabor_filter_direct (double sums, one rounding; anchor ks / 2, BORDER_REFLECT_101).  Sixteen copies of the box kernel stand in for the
    sixteen orientations: on an image scaled to [0, 1] the bank's clamp does nothing and the mean of sixteen equal planes is the plane."""
    src = _SRC_10683
    expected = np.array([
        83, 83, 77, 80, 76, 76, 76, 75, 71, 67, 72, 71, 73, 70, 80, 83, 86, 84, 89, 88, 88, 96, 99, 98,
        83, 83, 77, 80, 76, 76, 76, 75, 71, 67, 72, 71, 73, 70, 80, 83, 86, 84, 89, 88, 88, 96, 99, 98,
        82, 82, 77, 80, 77, 75, 74, 75, 70, 68, 71, 72, 72, 72, 82, 84, 88, 88, 93, 92, 93, 100, 105, 104,
        76, 76, 72, 77, 73, 74, 73, 74, 69, 68, 71, 71, 73, 72, 82, 81, 86, 87, 92, 91, 92, 98, 103, 102,
        75, 75, 72, 77, 73, 72, 75, 76, 74, 71, 73, 75, 76, 72, 81, 80, 85, 87, 90, 89, 90, 97, 102, 97,
        74, 74, 71, 77, 72, 74, 77, 76, 74, 72, 74, 76, 77, 76, 84, 83, 85, 87, 90, 92, 93, 100, 102, 99,
        72, 72, 69, 71, 68, 73, 73, 73, 70, 69, 74, 72, 75, 75, 81, 82, 85, 87, 90, 94, 96, 103, 102, 101,
        71, 71, 68, 70, 68, 71, 73, 71, 69, 68, 74, 72, 73, 73, 81, 80, 84, 89, 91, 99, 102, 107, 106, 105,
        74, 74, 70, 69, 67, 73, 76, 72, 69, 70, 79, 75, 74, 75, 82, 83, 88, 91, 92, 100, 104, 108, 106, 105,
        75, 75, 71, 70, 67, 75, 76, 71, 67, 68, 75, 72, 72, 75, 81, 83, 87, 89, 89, 97, 102, 107, 103, 103,
        69, 69, 67, 67, 65, 72, 74, 71, 70, 70, 75, 74, 74, 75, 80, 80, 84, 85, 85, 92, 96, 100, 97, 97,
        67, 67, 67, 68, 67, 77, 79, 75, 74, 76, 81, 78, 81, 80, 84, 81, 84, 83, 83, 91, 94, 95, 93, 93,
        73, 73, 71, 73, 70, 80, 82, 79, 80, 83, 85, 82, 82, 82, 87, 84, 88, 87, 84, 91, 93, 94, 93, 92,
        72, 72, 74, 75, 71, 80, 81, 79, 80, 82, 82, 80, 82, 84, 88, 83, 87, 87, 83, 88, 88, 89, 90, 90,
        78, 78, 81, 80, 74, 84, 86, 82, 85, 86, 85, 81, 83, 83, 86, 84, 85, 84, 78, 85, 82, 83, 85, 84,
        81, 81, 84, 81, 75, 86, 90, 85, 89, 91, 89, 84, 86, 87, 90, 87, 89, 85, 78, 84, 79, 80, 81, 81,
        76, 76, 80, 79, 73, 86, 90, 87, 92, 95, 92, 87, 91, 92, 93, 87, 89, 84, 77, 81, 76, 74, 76, 76,
        77, 77, 80, 77, 72, 83, 86, 86, 93, 95, 91, 87, 92, 92, 93, 87, 90, 84, 79, 79, 75, 72, 75, 72,
        80, 80, 81, 79, 72, 82, 86, 86, 95, 97, 89, 87, 89, 89, 91, 85, 88, 84, 79, 80, 73, 69, 74, 73,
        82, 82, 82, 80, 74, 83, 86, 87, 98, 100, 90, 90, 93, 94, 94, 89, 90, 84, 82, 79, 71, 68, 72, 69,
        76, 76, 77, 76, 70, 81, 83, 88, 99, 102, 92, 91, 97, 97, 97, 90, 90, 86, 83, 81, 70, 67, 70, 68,
        75, 75, 76, 74, 69, 79, 84, 88, 102, 106, 95, 94, 99, 98, 98, 90, 89, 86, 82, 79, 67, 62, 65, 62,
        80, 80, 82, 78, 71, 82, 87, 90, 105, 108, 96, 94, 99, 98, 97, 88, 88, 85, 81, 79, 65, 61, 65, 60,
        77, 77, 80, 75, 66, 76, 81, 87, 102, 105, 92, 91, 95, 97, 96, 88, 89, 88, 84, 81, 67, 63, 68, 63], np.int32).reshape(24, 24)
    ks = 12
    bank = np.full((16, ks, ks), np.float32(1.0) / np.float32(ks * ks), np.float32)
    plane = O.gabor_filter_direct(src / np.float32(255.0), ks, bank)
    got = np.rint(plane.astype(np.float64) * 255.0).astype(np.int32)
    assert np.abs(got - expected).max() <= 2, np.abs(got - expected).max()


def test_dft_filter2d_regression_13179():
    """Imgproc_Filter2D.dftFilter2d_regression_13179 (OCV/imgproc/test/test_filter.cpp:2223-2282): the 16 x 16 top-left ROI of a literal 24 x 24 8-bit image
    filtered with getGaborKernel(Size(13, 13), 8, 0, 3, 0.25) through filter2D's DFT path (the ROI is not isolated: its right and bottom borders are the
    parent image's pixels, its left and top ones the reflection), literal expected output, tolerance 2.  As in regression_10683 above the oracle's form is
    gabor_filter_direct with sixteen copies of the kernel on the image scaled to [0, 1] (the bank's clamp = the 8-bit saturation).  The kernel comes from
    getGaborKernel's formula (OCV/imgproc/src/gabor.cpp:50-95) restated here in double, AND from the oracle's own bank generator (orientation 0 of
    gabor_bank(13, 8, 3, 0.25, pi / 2) — the function the hot path's banks come from): the two must agree to float precision."""
    src = _SRC_10683
    expected = np.array([
        0, 255, 0, 0, 255, 0, 0, 255, 0, 0, 255, 255, 0, 255, 0, 0,
        0, 255, 0, 0, 255, 0, 0, 255, 0, 0, 255, 255, 0, 255, 0, 0,
        0, 255, 0, 0, 255, 0, 0, 255, 70, 0, 255, 255, 0, 255, 0, 0,
        0, 234, 138, 0, 255, 0, 0, 255, 8, 0, 255, 255, 0, 255, 0, 0,
        0, 0, 255, 0, 255, 228, 0, 255, 255, 0, 255, 255, 0, 255, 0, 5,
        0, 0, 255, 0, 255, 0, 0, 255, 0, 0, 255, 255, 0, 255, 0, 0,
        0, 253, 0, 0, 255, 0, 0, 255, 0, 0, 255, 255, 0, 255, 0, 0,
        0, 255, 0, 0, 255, 0, 0, 255, 0, 0, 255, 93, 0, 255, 0, 255,
        0, 255, 0, 0, 255, 0, 182, 255, 0, 0, 255, 0, 0, 255, 0, 0,
        0, 0, 253, 0, 228, 0, 255, 255, 0, 0, 255, 0, 0, 0, 0, 75,
        0, 0, 255, 0, 0, 0, 255, 255, 0, 255, 206, 0, 1, 162, 0, 255,
        0, 0, 255, 0, 0, 0, 255, 255, 0, 255, 255, 0, 0, 255, 0, 255,
        0, 0, 255, 0, 0, 0, 255, 255, 0, 255, 255, 0, 255, 255, 0, 255,
        0, 0, 255, 255, 0, 0, 255, 0, 0, 255, 255, 0, 255, 168, 0, 255,
        0, 0, 255, 255, 0, 0, 255, 26, 0, 255, 255, 0, 255, 255, 0, 255,
        0, 0, 255, 255, 0, 0, 255, 0, 0, 255, 255, 0, 255, 255, 0, 255], np.int32).reshape(16, 16)
    ks, sigma, theta, lambd, gamma, psi = 13, 8.0, 0.0, 3.0, 0.25, np.pi * 0.5
    r = ks // 2
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1].astype(np.float64)
    xr = xx * np.cos(theta) + yy * np.sin(theta); yr = -xx * np.sin(theta) + yy * np.cos(theta)
    v = np.exp(-0.5 / sigma ** 2 * xr ** 2 - 0.5 / (sigma / gamma) ** 2 * yr ** 2) * np.cos(2 * np.pi / lambd * xr + psi)
    kernel = np.ascontiguousarray(v[::-1, ::-1])                       # kernel(ymax - y, xmax - x) = v(x, y)
    own = O.gabor_bank(ks, sigma, lambd, gamma, psi)[0]
    assert np.abs(own.astype(np.float64) - kernel).max() < 1e-6
    for k32 in (kernel.astype(np.float32), own):
        bank = np.ascontiguousarray(np.broadcast_to(k32, (16, ks, ks)))
        plane = O.gabor_filter_direct(src / np.float32(255.0), ks, bank)
        got = np.rint(plane.astype(np.float64) * 255.0).astype(np.int32)[:16, :16]
        assert np.abs(got - expected).max() <= 2, np.abs(got - expected).max()


def test_gaussian_blur_regression_11303_property():
    """Imgproc_GaussianBlur.regression_11303 (OCV/imgproc/test/test_filter.cpp:2358-2367): a float GaussianBlur of a constant 2115 x 211 image must return the
    constant (norm L2 <= 1e-3) — normalised taps and reflected borders.  The reference call's kernel (sigma 8.64, 71 taps) is not on the path; the property is
    asserted for the oracle's own float Gaussians: the 9-tap sigma-1 blur inside unsharp_mask (src/util.cpp:113-148) and the 17-tap sigma-2 one of
    Extractor::keypoints' unsharp (src/extractor.cpp:57), at the regression's geometry."""
    img = np.ones((211, 2115, 3), np.float32)
    _, blur, _ = O.unsharp(img, 1.0, 0.3)
    assert float(np.sqrt(((blur.astype(np.float64) - 1.0) ** 2).sum())) <= 1e-3
    us = O.orb_unsharp_gray(np.full((211, 2115), 255, np.uint8))      # constant image: blurred == itself, the difference is 0, the output the input
    assert float(np.sqrt(((us.astype(np.float64) - 1.0) ** 2).sum())) <= 1e-3


def test_median_blur_hires_regression_13409():
    """Imgproc_MedianBlur.hires_regression_13409 (OCV/imgproc/test/test_filter.cpp:2284-2294): on an image with sides over 1024 (the reference tiles such images)
    medianBlur of a region equals the region of medianBlur away from the region's own border, ksize 9 — the first window of Extractor::foreground's chain
    (src/extractor.cpp:149).  The oracle's median has no tiles; the test pins that it is a pure sliding-window function."""
    rng = np.random.default_rng(13409)
    src = rng.integers(0, 256, (1100, 1300), dtype=np.uint8)
    whole = O.median_blur_u8(src, 9)
    part = O.median_blur_u8(np.ascontiguousarray(src[150:950, 200:1100]), 9)
    assert np.array_equal(whole[154:946, 204:1096], part[4:-4, 4:-4])
