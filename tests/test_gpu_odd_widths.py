"""Frames of widths that are not multiples of 4 (the reference's CLI pads every input to the UNION size of its images, src/poppy.cpp:186-239: any
width — 5 of its 26 sample images and 4 of its 16 demo pairs are 639 or 749 wide).  From 150 001 pixels up a level's rows are padded to a multiple
of 4 inside the frame slots (kernels.h: level_pitch), so that the fused raster + warp kernel, the wide pyramid kernels and the unsharp kernel's
16-byte loads take them: every intermediate and the frame against the oracle, bit for bit, on the fused path and on the id-map (debug) path."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from poppy_amd import capi, synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def _same(name, got, want):
    assert got.shape == want.shape, name
    neq = _bits(got) != _bits(want)
    if neq.any():
        idx = np.argwhere(neq)
        raise AssertionError(f"{name}: {len(idx)} of {got.size} elements differ, first at {idx[0]}, last at {idx[-1]}")


def _points(w, h, n, seed, spread):
    rng = np.random.default_rng(seed)
    p1 = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
    p2 = (p1 + rng.normal(0, spread, (n, 2))).astype(np.float32)
    p2[:, 0] = np.clip(p2[:, 0], 0, w - 1); p2[:, 1] = np.clip(p2[:, 1], 0, h - 1)
    corners = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]], np.float32)
    return np.concatenate([p1, corners]), np.concatenate([p2, corners])


# 749 x 480, 639 x 480: the reference's own odd sample widths (level 0 padded, level 1 tight); 1918 x 1080: levels 0 and 1 padded (959 is odd);
# 1001 x 700 and 1917 x 541: widths 1 and 3 more than a multiple of 4, odd heights; 398 x 377: just above the padding threshold; 397 x 377: just below
# it (tight rows everywhere: the id-map path)
SIZES = [(749, 480), (639, 480), (1918, 1080), (1001, 700), (1917, 541), (398, 377), (397, 377)]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("debug", [False, True])
def test_frame_stages_vs_oracle(w, h, debug):
    c = capi.Context(0)
    try:
        if debug:
            c.set_debug(True)
        p1, p2 = _points(w, h, 90, w + h, 6.0)
        c1 = synth.textured_bgr(w, h, 41); c2 = synth.textured_bgr(w, h, 42)
        g = synth.unit_field(w, h, 7)
        for s, m in ((0.3, 0.3), (0.75, 0.4)):
            want, wmp, d = O.morph_images(c1, c2, g, p1, p2, s, m, 64, debug=True)
            got, gmp = c.morph_images(c1, c2, g, p1, p2, s, m)
            if not debug and w * h > 150000:
                assert c.last_warp_kind() == 2, "the fused raster + warp kernel takes every width from 150 001 pixels up"
            for name in ("trImg1", "trImg2", "lbmask", "lapBlend"):
                _same(f"{name} {w}x{h} s={s}", c.fetch(name), d[name])
            _same(f"frame {w}x{h} s={s}", got, want)
    finally:
        c.close()


@pytest.mark.parametrize("w,h", [(749, 480), (1918, 1080)])
def test_chained_sequence_vs_oracle(w, h):
    """the default (chained) mode: every frame is the next one's first image; six frames, each against the oracle's chain"""
    c = capi.Context(0, number_of_frames=6)
    try:
        a, b = synth.gen_pair(w, h, 77)
        c.pair_begin(a, b)
        p1, p2 = c.pair_points()
        g = c.fetch("gabor2")
        frames = c.morph_frames(-1.0)
        cur, pts = a.copy(), p1.copy()
        for j in range(6):
            s = capi.lib().poppy_frame_ratio(j, 6, -1.0)
            want, mp, _ = O.morph_images(cur, b, g, pts, p2, s, s, 64, debug=True)
            _same(f"chained frame {j} {w}x{h}", frames[j], want)
            cur, pts = want, mp
    finally:
        c.close()


@pytest.mark.parametrize("case", ["a_639x480_numbers", "a_749x480_cars"])
def test_reference_demo_pairs_whole_morph(case):
    """The whole of poppy::morph on two of the reference's own demo pairs whose widths are no multiples of 4 (make_demos.sh:15 `cars`, 749 x 480; :31 `numbers`,
    639 x 480; pixels committed as tests/golden/demo_pairs.npz) against runs of the REAL reference on the same pixels: nfeatures, both details, the prepared
    point pairs, the printed morph distance, every chained frame (frame 0 in full, the others by sha256) and a phase-mode frame."""
    inp = G.astage_inputs(case)
    n = int(inp["cfg"][0])
    c = capi.Context(0, number_of_frames=n)
    rc, frames, dist = c.morph(inp["img1"], inp["img2"])
    assert rc == 0 and len(frames) == n
    nf, det = c.pair_begin_info()
    ref = G.full(case, "detail")
    assert nf == int(ref[3]) and det == (ref[0], ref[1])
    assert dist == float(G.full(case, "printedMorphDist")[0])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1)
    G.check(case, "prepared2", p2)
    assert c.last_warp_kind() == 2
    G.check(case, "frame0", frames[0])
    bad = [j for j, f in enumerate(frames) if G.sha(f) != G.entries(case)[f"frame{j}"]["sha256"]]
    assert not bad, f"frames {bad} differ from the reference"
    c.close()
    c1 = capi.Context(0, number_of_frames=1)                            # a phase-mode frame of the same pair
    rc, fr, _ = c1.morph(inp["img1"], inp["img2"], phase=float(inp["cfg"][4]))
    assert rc == 0 and len(fr) == 1
    assert G.sha(fr[0]) == G.entries(case)["phase0_frame"]["sha256"]
    c1.close()


def test_streaming_unsharp_on_odd_widths():
    """k_unsharp_stream takes any width and source pitch since round 6 (4K-class odd widths ran on the tile kernel before: 3838 x 2160 +10 % per frame): the row's
    last, shorter pixel group goes out byte by byte, the other groups' dword stores are unaligned.  The stage and chained tests of this file once more with the
    streaming kernel forced for every size (POPPY_UNSHARP_STREAM is read once per process), and one 4K-class odd width, where it is the kernel launch_unsharp
    picks by itself, against the oracle."""
    import os
    import subprocess
    import sys
    if os.environ.get("POPPY_UNSHARP_STREAM") or os.environ.get("POPPY_UNSHARP_TILE"):
        pytest.skip("a kernel form is already forced in this process")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "frame_stages or chained_sequence"],
                       env=dict(os.environ, POPPY_UNSHARP_STREAM="1"), capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    w, h = 2878, 1618                      # 4.66 Mpx: the streaming kernel by launch_unsharp's own choice; width 2 more than a multiple of 4
    c = capi.Context(0)
    try:
        p1, p2 = _points(w, h, 120, 77, 9.0)
        c1 = synth.textured_bgr(w, h, 43); c2 = synth.textured_bgr(w, h, 44)
        g = synth.unit_field(w, h, 9)
        want, _ = O.morph_images(c1, c2, g, p1, p2, 0.4, 0.4, 64)
        got, _ = c.morph_images(c1, c2, g, p1, p2, 0.4, 0.4)
        assert c.last_warp_kind() == 2
        _same(f"frame {w}x{h}", got, want)
    finally:
        c.close()
