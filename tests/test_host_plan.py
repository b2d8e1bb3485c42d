"""CPU tests of the product's host side: C-ABI surface and mesh planning (no GPU needed)."""
import re
import os
import numpy as np
import pytest

import golden_util as G
from poppy_amd import capi

ROOT = G.ROOT


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "poppy_hip.h")).read()
    declared = set(re.findall(r"\b(poppy_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"poppy_write_cb"}
    L = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libpoppy_hip.so lacks {missing}"
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))


def test_create_fails_loudly_without_gpu(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(capi.PoppyError):
        capi.Context(0)


@pytest.mark.parametrize("case", ["b_64x48", "b_256x256", "b_509x381", "b_1920x1080"])
def test_plan_frame_matches_reference(case):
    inp = G.bstage_inputs(case)
    w, h, n, ratios, levels = G.make_inputs.BSTAGE[case]
    for k, (s, m) in enumerate(ratios):
        pf = f"f{k}_"
        plan = capi.plan_frame(w, h, inp["pts1"], inp["pts2"], s)
        G.check(case, pf + "morphedPoints", plan["morphed"])
        G.check(case, pf + "triIdx", plan["idx3"])
        G.check(case, pf + "triMorphInt", plan["tri_xy"])
        G.check(case, pf + "M1", plan["M1"])
        G.check(case, pf + "M2", plan["M2"])


def test_plan_frame_out_of_range_point():
    p = np.array([[10, 10], [64, 20], [30, 40]], np.float32)     # x == cols survives clip_points and must fail
    with pytest.raises(capi.PoppyError):
        capi.plan_frame(64, 48, p, p, 0.5)


def test_frame_ratio_scheduler():
    L = capi.lib()
    N = 60
    assert L.poppy_frame_ratio(0, N, -1.0) == 0.0
    for j in range(1, N):
        lin = j / float(N)
        assert L.poppy_frame_ratio(j, N, -1.0) == min(1.0, (1.0 / (1.0 - lin)) / N)
    assert L.poppy_frame_ratio(0, 1, 0.5) == 0.5           # --frames 1 --phase t  =>  shape = t
    assert L.poppy_frame_ratio(0, 60, 0.5) == 0.5 * (1.0 / 60)


@pytest.mark.parametrize("case", ["m_640x480", "m_1920x1080", "m_tol2"])
def test_host_point_matcher_matches_reference(case):
    """poppy_match_points = filter_invalid_points + morph_distance + Matcher::match/prepare (host C++, no GPU)."""
    inp = G.match_inputs(case)
    w, h, tol = int(inp["cfg"][0]), int(inp["cfg"][1]), float(inp["cfg"][2])
    a, b, imd = capi.match_points(inp["pts1"], inp["pts2"], w, h, tol)
    G.check(case, "initialMorphDist", np.array([imd]))
    G.check(case, "prepared1", a)
    G.check(case, "prepared2", b)


def test_host_point_matcher_on_real_keypoints():
    case = "a_512x384_chain"
    kp1, kp2 = G.full(case, "kp1"), G.full(case, "kp2")
    n = min(len(kp1), len(kp2))
    a, b, imd = capi.match_points(kp1[:n, :2], kp2[:n, :2], 512, 384, 1.0)
    G.check(case, "initialMorphDist", np.array([imd]))
    G.check(case, "prepared1", a)
    G.check(case, "prepared2", b)


@pytest.mark.parametrize("seed", range(8))
def test_host_point_matcher_ties_and_tombstones_against_the_oracle(seed):
    """The host matcher picks partners by SQUARED distance and takes a root only where floats could tie (point_match.cpp: greedy_pairs);
    the oracle evaluates the reference's hypot for every candidate (oracle/match.cpp: make_distance_map, src/util.cpp:351-383).  Point sets made
    to tie: integer grids, repeated points, ORB-style coordinates (integers times 1.2^level), the reference's tombstone value (-1, -1) —
    filter_invalid_points keeps nothing negative, so that one enters through the pairing alone, below."""
    import oracle_lib as O
    rng = np.random.default_rng(100 + seed)
    w, h, n = 640, 480, 300
    kind = seed % 4
    if kind == 0:                                          # coarse integer grid: many equal distances
        p1 = rng.integers(0, 40, (n, 2)).astype(np.float32) * 12
        p2 = rng.integers(0, 40, (n, 2)).astype(np.float32) * 12
    elif kind == 1:                                        # ORB-like: level coordinates scaled by 1.2^level
        lv = rng.integers(0, 8, n)
        sc = (np.float32(1.2) ** lv).astype(np.float32)
        p1 = (rng.integers(16, 300, (n, 2)).astype(np.float32) * sc[:, None]).astype(np.float32)
        p2 = (rng.integers(16, 300, (n, 2)).astype(np.float32) * sc[::-1, None]).astype(np.float32)
    elif kind == 2:                                        # duplicates of a few points, both sets
        base = (rng.random((20, 2)) * [w - 1, h - 1]).astype(np.float32)
        p1 = base[rng.integers(0, 20, n)]; p2 = base[rng.integers(0, 20, n)]
    else:                                                  # near-ties: a jitter of one float ulp around common positions
        base = (rng.random((30, 2)) * [w - 1, h - 1]).astype(np.float32)
        p1 = base[rng.integers(0, 30, n)]
        p2 = np.nextafter(base[rng.integers(0, 30, n)], np.float32(1e9) * rng.choice([-1, 1], (n, 2)).astype(np.float32)).astype(np.float32)
    p1 = np.clip(p1, 0, [w, h]).astype(np.float32); p2 = np.clip(p2, 0, [w, h]).astype(np.float32)
    f1, f2 = O.filter_invalid(p1, p2, w, h)
    imd = O.morph_distance(f1, f2, w, h)
    e1, e2 = O.match_prepare(f1, f2, w, h, 1.0, imd)
    a, b, got = capi.match_points(p1, p2, w, h, 1.0)
    assert got == imd
    assert np.array_equal(a, e1) and np.array_equal(b, e2)


def test_match_points_all_out_of_image():
    p = np.array([[-5, 3], [700, 2]], np.float32)
    a, b, imd = capi.match_points(p, p, 64, 48, 1.0)
    assert len(a) == 0 and len(b) == 0


def test_warp_records_layout_and_admission():
    """Records of the tiled warp kernel (poppy_amd/csrc/frame_plan.h): layout, identity record, the zero-matrix
    substitution of create_map's z == 0 case (src/algo.cpp:166-167) and the admission rule."""
    from poppy_amd import synth
    w, h = 640, 480
    p1, p2 = synth.point_pairs(w, h, 60, seed=3, dup=0, oob=0)
    plan = capi.plan_frame(w, h, p1, p2, 0.4)
    i1 = plan["inv1"].reshape(-1, 9); i2 = plan["inv2"].reshape(-1, 9)
    rec, ok = capi.warp_records(i1, i2, w, h)
    assert ok and rec.shape == (len(i1) + 1, 20)
    ident = np.zeros(20, np.float32); ident[[0, 3, 6, 9]] = 32; ident[[16, 17]] = 1
    assert np.array_equal(rec[0], ident)
    # a = inv1, b = inv2: 32 x {h0,h3}a {h1,h4}a {h2,h5}a {h0,h3}b {h1,h4}b {h2,h5}b, then {h6a,h6b} {h7a,h7b} {h8a,h8b} pad pad
    # (the numerator rows carry remap's sub-pixel scale of 32: an exact scaling, round 4)
    order_a = [0, 3, 1, 4, 2, 5]
    assert np.array_equal(rec[1:, 0:6], np.float32(32) * i1[:, order_a]) and np.array_equal(rec[1:, 6:12], np.float32(32) * i2[:, order_a])
    assert np.array_equal(rec[1:, 12:18:2], i1[:, 6:9]) and np.array_equal(rec[1:, 13:18:2], i2[:, 6:9])
    assert not rec[:, 18:].any()

    one = np.array([[1, 0, 0, 0, 1, 0, 0, 0, 1]], np.float32)
    zero = np.zeros((1, 9), np.float32)
    rec, ok = capi.warp_records(zero, one, w, h)              # singular triangle: all-zero inverse, z := 1e-5
    assert ok and rec[1, 16] == np.float32(0.00001) and rec[1, 17] == 1 and not rec[1, 0:6].any()
    assert not capi.warp_records(np.asarray([[1e-35, 0, 0, 0, 1, 0, 0, 0, 1]], np.float32), one, w, h)[1]    # a numerator entry whose products could be subnormal

    def admitted(m):
        return capi.warp_records(np.asarray([m], np.float32), one, w, h)[1]
    assert admitted([1, 0, 0, 0, 1, 0, 1e-4, 0, 1])           # denominator stays in [1, 1.064]
    assert not admitted([1, 0, 0, 0, 1, 0, -2e-3, 0, 1])      # 1 - 0.002 * 639 changes sign across the image
    assert not admitted([1, 0, 0, 0, 1, 0, 0, 0, 0])          # z == 0 everywhere with a non-zero matrix
    assert not admitted([1, 0, 0, 0, 1, 0, 0, 0, 1e-7])       # below 2^-20
    assert not admitted([1, 0, 0, 0, 1, 0, 0, 0, 2e6])        # above 2^20
    assert admitted([1, 0, 0, 0, 1, 0, 0, 0, -1])             # negative denominators are fine while they keep their sign
    assert not admitted([np.inf, 0, 0, 0, 1, 0, 0, 0, 1]) and not admitted([np.nan, 0, 0, 0, 1, 0, 0, 0, 1])
    assert not admitted([2e12, 0, 0, 0, 1, 0, 0, 0, 1])       # > 2^40


def test_ratio_symmetry_host_vs_reference_lists():
    """poppy_ratio_symmetry (host) on the reference's own 2-NN lists reproduces its symmetric matches."""
    for case, nf in (("o_256x256", 300), ("o_640x480", 500), ("o_1920x1080", 516)):
        k12, k21 = G.full(case, f"n{nf}_knn12"), G.full(case, f"n{nf}_knn21")
        G.check(case, f"n{nf}_sym", capi.ratio_symmetry(k12, k21, 0.7))
    # no second neighbour -> dropped; 0/0 -> kept; ratio 0 keeps only exact matches
    k12 = np.array([[0, 5, -1, -1], [1, 0, 0, 0]], np.int32); k21 = np.array([[0, 5, 1, 9], [1, 0, 0, 0]], np.int32)
    assert capi.ratio_symmetry(k12, k21, 0.7).tolist() == [[1, 1, 0]]
    assert len(capi.ratio_symmetry(k12[:0], k21, 0.7)) == 0


@pytest.mark.parametrize("case", ["l_317x211", "l_320x240", "l_640x480"])
def test_procrustes_and_perspective_fit_host(case):
    """Host math of the auto-align (poppy_amd/csrc/auto_align.cpp) against the reference's Procrustes and OpenCV's
    getPerspectiveTransform (fixtures l_*; no GPU)."""
    inp = G.make_inputs.align_inputs(case)
    r = capi.procrustes(inp["pts1"], inp["pts2"])
    G.check(case, "pc_rotation", r["rotation"])
    sc = G.full(case, "pc_scalars")
    assert np.float32(sc[0]) == r["scale"] and np.float32(sc[1]) == r["error"]
    G.check(case, "pc_yprime", r["yprime"])
    G.check(case, "prim_persp", capi.perspective_from4(inp["pts1"][:4], inp["pts2"][:4]))


def test_dft_plan_tables():
    """The transform plan of dft_detail2's kernels (pass order, load permutation, float twiddles): same tables as the
    implementation that reproduced cv::dft bit for bit (tests/golden/dft_plan_fnv.json), and the same permutation /
    twiddles as the oracle's independent planner implies through dft_detail2 (tests/test_oracle_sequences.py)."""
    import json, os
    ref = json.load(open(os.path.join(G.GOLD, "dft_plan_fnv.json")))["fnv1a64"]

    def fnv(chunks):
        h = 1469598103934665603
        for c in chunks:
            for b in c.tobytes():
                h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h
    for n_s, want in ref.items():
        n = int(n_s)
        f, itab, wave = capi.dft_plan(n)
        parts = [np.array([len(f)], np.int32), f, itab] + ([wave] if (n > 5 or n == 4) else [])
        assert fnv(parts) == want, n
        assert sorted(itab.tolist()) == list(range(n))                     # a permutation
        assert int(np.prod(f.astype(np.int64))) == max(n, 1)
    f, itab, wave = capi.dft_plan(1080)
    assert f.tolist() == [8, 5, 3, 3, 3]                                    # power of two, then the odd primes descending
    assert capi.dft_plan(1920)[0].tolist() == [128, 5, 3]
    k = np.arange(1080)
    assert np.abs(wave[:, 0] - np.cos(2 * np.pi * k / 1080)).max() < 1e-6 and np.abs(wave[:, 1] + np.sin(2 * np.pi * k / 1080)).max() < 1e-6


def test_matcher_hypot_is_libm_hypotf():
    """point_match.cpp evaluates hypotf by glibc's own closed form; compare with the linked libm on 2e6 inputs."""
    L = capi.lib()
    assert L.poppy_hypotf_selfcheck(2000000, 12345) == 0
    assert L.poppy_hypotf_selfcheck(500000, 999) == 0


def test_shim_header_compiles_against_reference_headers():
    """include/poppy_hip_shim.hpp, the binding a Poppy maintainer adds, syntax-checked against the vendored OpenCV headers and
    Poppy's settings.hpp (only where the reference tree exists: this container)."""
    import subprocess
    if not (os.path.isdir("/root/reference/src") and os.path.exists("/tmp/ocv-build/opencv2/opencv_modules.hpp")):
        pytest.skip("reference tree / OpenCV build tree not present")
    subprocess.check_call([os.path.join(ROOT, "tools", "check_shim.sh")])


def test_file_sinks(tmp_path):
    """poppy_sink_* (raw / PPM / Y4M writers with the poppy_write_cb signature): written files parse back to the frames."""
    from poppy_amd import synth
    L = capi.lib()
    w, h = 37, 21
    frames = [synth.textured_bgr(w, h, 5 + k) for k in range(3)]
    padded = [np.ascontiguousarray(np.pad(f, ((0, 0), (0, 5), (0, 0)))) for f in frames]          # a row stride larger than w*3
    raw = tmp_path / "out.bgr"; ppm = str(tmp_path / "f%03d.ppm"); y4m = tmp_path / "out.y4m"
    for path, fmt in ((str(raw), 0), (ppm, 1), (str(y4m), 2)):
        s = L.poppy_sink_open(path.encode(), fmt, w, h, 30, 1)
        assert s
        for f in padded:
            L.poppy_sink_write(s, f.ctypes.data, w, h, f.strides[0])
        assert L.poppy_sink_close(s) == 3
    assert np.array_equal(np.fromfile(raw, np.uint8).reshape(3, h, w, 3), np.stack(frames))
    for k in range(3):
        data = open(ppm % k, "rb").read()
        head = b"P6\n%d %d\n255\n" % (w, h)
        assert data.startswith(head)
        rgb = np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3)
        assert np.array_equal(rgb[..., ::-1], frames[k])
    data = open(y4m, "rb").read()
    assert data.startswith(b"YUV4MPEG2 W37 H21 F30:1") and data.count(b"FRAME\n") == 3
    body = data[data.index(b"\n") + 1:]
    yplane = np.frombuffer(body[6:6 + w * h], np.uint8).reshape(h, w)
    f0 = frames[0].astype(np.int64)
    assert np.array_equal(yplane, (19595 * f0[..., 2] + 38470 * f0[..., 1] + 7471 * f0[..., 0] + 32768) >> 16)
    # a frame of another geometry poisons the sink
    s = L.poppy_sink_open(str(raw).encode(), 0, w, h, 30, 1)
    L.poppy_sink_write(s, padded[0].ctypes.data, w + 1, h, padded[0].strides[0])
    assert L.poppy_sink_close(s) < 0


def test_ppm_sink_pattern_is_parsed_not_printed(tmp_path):
    """The PPM path is never handed to printf: exactly one %d / %<w>d / %0<w>d is substituted by the library, "%%" is a literal percent
    sign, everything else (no conversion, a second one, %s, %n, %x) is refused at open."""
    L = capi.lib()
    for bad in ("plain.ppm", "a%d_%d.ppm", "x%s.ppm", "x%n.ppm", "x%x.ppm", "x%", "x%5", "x%999999999d.ppm"):
        assert not L.poppy_sink_open(str(tmp_path / bad).encode(), 1, 4, 4, 30, 1), bad
    f = np.zeros((4, 4, 3), np.uint8)
    for pat, name in (("p%d.ppm", "p0.ppm"), ("q%4d.ppm", "q   0.ppm"), ("r%03d_100%%.ppm", "r000_100%.ppm")):
        s = L.poppy_sink_open(str(tmp_path / pat).encode(), 1, 4, 4, 30, 1)
        assert s, pat
        L.poppy_sink_write(s, f.ctypes.data, 4, 4, f.strides[0])
        assert L.poppy_sink_close(s) == 1
        assert (tmp_path / name).exists(), name
