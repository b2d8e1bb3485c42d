"""Host-side tables behind two device kernels, checked without a GPU: the paired kernel spectra of the FFT form of the Gabor banks
(kernels_gabor_fft.hip) against numpy's FFT, and the tap table of the pyramid tail (kernels_pyramid_tail.hip) against an independent
restatement of the reference's addressing (OCV/imgproc/src/pyramids.cpp:745-1005, OCV/core/src/copy.cpp:748-792)."""
import numpy as np
import pytest


def perm64(p):
    return (p >> 3) + 8 * (p & 7)


@pytest.mark.parametrize("ks", [31, 13])
def test_gabor_fft_tables_vs_numpy(ks):
    from poppy_amd import capi
    bank, spec = capi.gabor_tables(ks)
    assert bank.shape == (16, ks, ks) and spec.shape == (8, 64, 64)
    pad = np.zeros((16, 64, 64))
    pad[:, :ks, :ks] = bank.astype(np.float64)
    K = np.fft.fft2(pad)
    pi = np.array([perm64(p) for p in range(64)])
    for j in range(8):
        # correlation = IFFT(patch^ . conj(K^)); a pair of real planes rides as real / imaginary part; 1 / 4096 = the inverse's scale
        G = (np.conj(K[2 * j]) + 1j * np.conj(K[2 * j + 1])) / 4096.0
        want = G[np.ix_(pi, pi)]                     # position (r, c) of the transform holds frequency (perm(r), perm(c))
        assert np.abs(want - spec[j]).max() <= 4e-16 * np.abs(G).max()
    # the bank itself: cv::getGaborKernel's centre tap is cos(psi) = cos(pi / 4) for every orientation
    assert np.all(bank[:, ks // 2, ks // 2] == np.float32(np.cos(np.pi / 4)))


def reflect101(p, n):
    if 0 <= p < n:
        return p
    if n == 1:
        return 0
    while not (0 <= p < n):
        p = -p if p < 0 else 2 * n - 2 - p
    return p


def down_geom(sw, sh, cn):
    dw = (sw + 1) // 2
    w0 = min(int((sw - 3) / 2) + 1, dw)              # C division truncates toward zero
    width = w0 * cn - cn
    covered = 0
    if width >= 4:
        covered = ((width - 4) // 4 + 1) * 4 if cn == 1 else ((width - 4) // 3 + 1) * 3
    return dw, (sh + 1) // 2, cn + covered, (dw * cn // 4) * 4


@pytest.mark.parametrize("w,h,levels,tail_px", [(1920, 1080, 64, 600), (3840, 2160, 64, 600), (640, 480, 64, 600), (97, 61, 64, 600),
                                                (1000, 37, 12, 600), (33, 150, 5, 600), (512, 512, 64, 2000), (8, 6, 64, 600)])
def test_pyr_tail_plan_matches_the_reference_addressing(w, h, levels, tail_px):
    from poppy_amd import capi
    info, desc = capi.pyr_tail_plan(w, h, levels, tail_px)
    lv = [(w, h)]
    for _ in range(levels):
        lv.append(((lv[-1][0] + 1) // 2, (lv[-1][1] + 1) // 2))
    first = next((i for i in range(1, levels + 1) if lv[i][0] * lv[i][1] <= tail_px), levels)
    assert info["first"] == first
    k1 = next((i for i in range(first, levels + 1) if lv[i] == (1, 1)), levels)
    assert info["wide_steps"] == k1 - first
    assert info["single_pixel_reductions"] == (levels - k1 if lv[k1] == (1, 1) else -1)
    if not info["ok"]:
        return
    at = 0
    for k in range(info["wide_steps"]):                      # pyrDown steps: 3-channel descriptors (L and R share them), then the mask's
        (sw, sh), (dw_, dh_) = lv[first + k], lv[first + k + 1]
        for cn in (3, 1):
            dw, dh, hBodyEnd, vBodyEnd = down_geom(sw, sh, cn)
            assert (dw, dh) == (dw_, dh_)
            for y in range(dh):
                for xe in range(dw * cn):
                    d = desc[at]; at += 1
                    px, c = divmod(xe, cn)
                    rows = [reflect101(2 * y + t - 2, sh) for t in range(5)]
                    cols = [reflect101(2 * px + t - 2, sw) * cn + c for t in range(5)]
                    got_rows = [int(d[0]) & 1023, (int(d[0]) >> 10) & 1023, (int(d[0]) >> 20) & 1023, int(d[1]) & 1023, (int(d[1]) >> 10) & 1023]
                    c2 = int(d[1]) >> 20
                    rel = [((int(d[2]) >> s) & 255) - 256 * (((int(d[2]) >> s) & 255) >> 7) for s in (0, 8, 16, 24)]     # signed bytes
                    got_cols = [c2 + int(rel[0]), c2 + int(rel[1]), c2, c2 + int(rel[2]), c2 + int(rel[3])]
                    assert got_rows == rows and got_cols == cols, (k, cn, y, xe)
                    assert ((int(d[0]) >> 30) & 1) == int(xe >= cn and xe < hBodyEnd) and (int(d[0]) >> 31) == int(xe < vBodyEnd), (k, cn, y, xe)
    for k in range(info["wide_steps"]):                      # collapse steps: pyrUp taps of every output element
        (cw, ch_), (nw, nh) = lv[first + k], lv[first + k + 1]
        for y in range(ch_):
            for xe in range(cw * 3):
                d = desc[at]; at += 1
                dpx, c = divmod(xe, 3)
                spx, oddx, sy, oddy = dpx >> 1, dpx & 1, y >> 1, y & 1
                if nw == 1:
                    form, cm, c0, cp = 4, 0, 0, 0
                elif spx == 0:
                    form, cm, c0, cp = (1 if oddx else 2), 0, 0, 1
                elif spx >= nw - 1:
                    form, cm, c0, cp = (4 if oddx else 3), nw - 2, nw - 1, spx
                else:
                    form, cm, c0, cp = (1 if oddx else 0), spx - 1, spx, spx + 1
                y1 = sy
                y2 = reflect101((sy + 1) * 2, nh * 2) >> 1
                y0 = y1 if oddy else reflect101((sy - 1) * 2, nh * 2) >> 1
                got = dict(y0=int(d[0]) & 1023, y1=(int(d[0]) >> 10) & 1023, y2=(int(d[0]) >> 20) & 1023, oddy=(int(d[0]) >> 30) & 1,
                           cm=int(d[1]) & 2047, c0=(int(d[1]) >> 11) & 2047, form=int(d[1]) >> 22, cp=int(d[2]) & 2047, mpx=int(d[2]) >> 11)
                assert got["form"] == form and got["oddy"] == oddy and (got["y0"], got["y1"], got["y2"]) == (y0, y1, y2), (k, y, xe, got)
                assert got["c0"] == c0 * 3 + c and got["mpx"] == y * cw + dpx, (k, y, xe, got)
                if form in (0, 3):
                    assert got["cm"] == cm * 3 + c
                if form in (0, 1, 2):
                    assert got["cp"] == cp * 3 + c
    assert at == info["descriptors"]


@pytest.mark.parametrize("w,h", [(160, 120), (317, 211), (64, 49), (33, 150)])
def test_radial_mask_table_vs_oracle_and_known_answers(w, h):
    """The mask Settings::enable_radial_mask multiplies into Extractor::foreground (draw_radial_gradiant, src/draw.cpp:21-38; the
    conversion of src/extractor.cpp:181-183): the library's host table against the oracle's restatement, and what the formula says
    without running it: 1 at the centre pixel, 0 at the pixel farthest from it, multiples of 1/255, symmetric about the centre."""
    import oracle_lib as O
    from poppy_amd import capi
    a = capi.radial_mask(w, h)
    b = O.radial_mask(w, h)
    assert a.dtype == np.float32 and a.shape == (h, w)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    cx, cy = int(w / 2.0), int(h / 2.0)
    assert a[cy, cx] == 1.0 and a.min() == 0.0 and a.max() == 1.0
    k = np.rint(a.astype(np.float64) * 255.0)
    assert np.array_equal((k.astype(np.float32) * np.float32(1.0 / 255.0)), a)
    d2 = (np.arange(w)[None, :] - cx) ** 2 + (np.arange(h)[:, None] - cy) ** 2
    assert a.ravel()[np.argmax(d2)] == 0.0
    n = min(cx, w - 1 - cx)
    assert np.array_equal(a[:, cx - n:cx][:, ::-1], a[:, cx + 1:cx + 1 + n])       # mirror columns about the centre column


def test_oracle_foreground_radial_mask_only_lowers_the_mask():
    """With the radial mask on, the log-curve input is fgMask * radial <= fgMask, so finalMask can only go down, pixel by pixel."""
    import oracle_lib as O
    from poppy_amd import synth
    img = synth.textured_bgr(96, 64, 9)
    plain = O.foreground(img)
    O.set_radial_mask(True)
    try:
        rad = O.foreground(img)
    finally:
        O.set_radial_mask(False)
    assert np.array_equal(rad["acc12"], plain["acc12"]) and np.array_equal(rad["blur12"], plain["blur12"])
    assert (rad["finalMask"] <= plain["finalMask"]).all() and (rad["finalMask"] < plain["finalMask"]).any()
    assert np.array_equal(O.foreground(img)["foreground"], plain["foreground"])    # the switch is off again


@pytest.mark.parametrize("case", ["a_256x256_radial", "a_320x200_radial"])
def test_radial_mask_against_the_reference_run(case):
    """Round 4: the option is pinned by a run of the real reference with Settings::enable_radial_mask (oracle/golden_gen: flags bit 1): the mask
    table of the library and of the oracle, and the oracle's foregrounds of both images, against the fixture — bit for bit."""
    import golden_util as G
    import oracle_lib as O
    from poppy_amd import capi
    inp = G.astage_inputs(case)
    h, w = inp["img1"].shape[:2]
    G.check(case, "radialMask", capi.radial_mask(w, h))
    G.check(case, "radialMask", O.radial_mask(w, h))
    O.set_radial_mask(True)
    try:
        G.check(case, "goodFeatures1", O.foreground(inp["img1"])["foreground"])
        G.check(case, "goodFeatures2", O.foreground(inp["img2"])["foreground"])
    finally:
        O.set_radial_mask(False)
