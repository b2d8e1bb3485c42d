"""GPU parity of the per-frame path: HIP kernels (through the C ABI) vs the oracle and the reference fixtures.

Bar: bit-exact on every integer/byte stage (triMap, warped images, final frame) and — because the kernels
reproduce the reference's float association and are built without FMA contraction — bit-exact on the float
stages as well (lbmask, lapBlend, unsharp).  north_star allows 1 LSB on blended pixels; we assert 0.
"""
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from poppy_amd import capi, synth

pytestmark = pytest.mark.gpu
TILED = 0 if os.environ.get("POPPY_HIP_GENERALWARP") is not None else 1    # the switch forces the general warp kernel


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    c.set_debug(True)
    yield c
    c.close()


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def _same(name, got, want):
    assert got.shape == want.shape, name
    neq = _bits(got) != _bits(want)
    if neq.any():
        idx = np.argwhere(neq)
        raise AssertionError(f"{name}: {len(idx)} of {got.size} elements differ, first at {idx[0]}: {got[tuple(idx[0])]} vs {want[tuple(idx[0])]}")


@pytest.mark.parametrize("case", ["b_64x48", "b_256x256", "b_509x381"])
def test_frame_vs_reference_fixtures(ctx, case):
    inp = G.bstage_inputs(case)
    w, h, n, ratios, levels = G.make_inputs.BSTAGE[case]
    for k, (s, m) in enumerate(ratios):
        pf = f"f{k}_"
        out, mp = ctx.morph_images(inp["c1"], inp["c2"], inp["gabor2"], inp["pts1"], inp["pts2"], s, m)
        G.check(case, pf + "morphedPoints", mp)
        for name in ("triMap", "trImg1", "trImg2", "lbmask", "lapBlend", "unsharp"):
            G.check(case, pf + name, ctx.fetch(name), what=f"ratio {s}")
        G.check(case, pf + "frame", out)


def test_frame_1080p_vs_reference_and_oracle(ctx):
    case = "b_1920x1080"
    inp = G.bstage_inputs(case)
    s, m = G.make_inputs.BSTAGE[case][3][0]
    out, mp = ctx.morph_images(inp["c1"], inp["c2"], inp["gabor2"], inp["pts1"], inp["pts2"], s, m)
    for name in ("triMap", "trImg1", "trImg2", "lbmask", "lapBlend", "unsharp"):
        G.check(case, "f0_" + name, ctx.fetch(name))
    G.check(case, "f0_frame", out)


@pytest.mark.parametrize("w,h,n", [(97, 61, 9), (200, 33, 20), (33, 200, 20), (640, 360, 120)])
def test_frame_vs_oracle_ragged_sizes(ctx, w, h, n):
    c1 = synth.textured_bgr(w, h, 51); c2 = synth.textured_bgr(w, h, 52)
    g = synth.unit_field(w, h, 13)
    p1, p2 = synth.point_pairs(w, h, n, seed=w + h)
    for s in (0.0, 0.37, 1.0):
        want, wmp, d = O.morph_images(c1, c2, g, p1, p2, s, s, 64, debug=True)
        got, gmp = ctx.morph_images(c1, c2, g, p1, p2, s, s)
        _same("morphed", gmp, wmp)
        for name in ("triMap", "trImg1", "trImg2", "lbmask", "lapBlend", "unsharp"):
            _same(f"{name} {w}x{h} s={s}", ctx.fetch(name), d[name])
        _same("frame", got, want)


def test_chained_sequence_vs_oracle(ctx):
    """Default CLI mode (src/poppy.hpp:177-219): frame j warps frame j-1 and the previous morphed points."""
    w, h, n, N = 320, 200, 40, 6
    c1 = synth.textured_bgr(w, h, 61); c2 = synth.textured_bgr(w, h, 62)
    g = synth.unit_field(w, h, 14)
    p1, p2 = synth.point_pairs(w, h, n, seed=9, dup=0, oob=0)
    ctx.pair_load(c1, c2, g, p1, p2)
    cur, pts = c1, p1
    L = capi.lib()
    for j in range(N):
        s = L.poppy_frame_ratio(j, N, -1.0)
        want, mp = O.morph_images(cur, c2, g, pts, p2, s, s, 64)
        got = ctx.render(s, s, chain=True)
        _same(f"chained frame {j}", got, want)
        cur, pts = want, mp


def test_morph_frames_callback_and_reset(ctx):
    w, h, n = 160, 120, 16
    c1 = synth.textured_bgr(w, h, 71); c2 = synth.textured_bgr(w, h, 72)
    g = synth.unit_field(w, h, 15)
    p1, p2 = synth.point_pairs(w, h, n, seed=3, dup=0, oob=0)
    c = capi.Context(0, number_of_frames=4)
    c.pair_load(c1, c2, g, p1, p2)
    a = c.morph_frames(-1.0)
    assert len(a) == 4
    c.reset()
    b = c.morph_frames(-1.0)
    for x, y in zip(a, b):
        assert (x == y).all()
    c.reset()
    one = c.morph_frames(0.5)          # phase mode: exactly one frame (src/poppy.hpp:234-235)
    assert len(one) == 1
    want, _ = O.morph_images(c1, c2, g, p1, p2, 0.5 / 4, 0.5 / 4, 64)
    _same("phase frame", one[0], want)
    c.close()


def test_no_points_and_dissolve(ctx):
    w, h = 64, 40
    a = synth.textured_bgr(w, h, 81); b = synth.textured_bgr(w, h, 82)
    g = synth.unit_field(w, h, 16)
    empty = np.zeros((0, 2), np.float32)
    with pytest.raises(capi.PoppyError):
        ctx.morph_images(a, b, g, empty, empty, 0.5, 0.5)
    for phase in (0.25, 0.5, 1.0 / 3):
        got = ctx.dissolve(a, b, phase)
        fa, fb = np.float32(phase), np.float32(1.0 - phase)
        want = np.clip(np.rint(b.astype(np.float32) * fa + a.astype(np.float32) * fb), 0, 255).astype(np.uint8)
        _same("dissolve", got, want)


def test_full_size_round_trip_properties(ctx):
    """Size-independent properties at BASELINE.json's 1080p size: shape 0 with mask 0 keeps image 1's warp the
    identity; shape 1 / mask 1 maps onto image 2; rendering is deterministic and chain reset is exact."""
    w, h = 1920, 1080
    c1, c2 = synth.gen_pair(w, h)
    g = synth.unit_field(w, h, 17)
    p1, p2 = synth.point_pairs(w, h, 300, seed=21, dup=0, oob=0)
    ctx.pair_load(c1, c2, g, p1, p2)
    f0 = ctx.render(0.0, 0.0)
    assert (ctx.fetch("trImg1") == c1).all()          # identity warp of source 1 at shape 0
    f1 = ctx.render(1.0, 1.0)
    assert (ctx.fetch("trImg2") == c2).all()          # identity warp of source 2 at shape 1
    a = ctx.render(0.4, 0.4); b = ctx.render(0.4, 0.4)
    assert (a == b).all()
    tm = ctx.fetch("triMap")
    assert tm.min() >= 0 and tm.max() <= len(ctx.triangles()[0])


def test_frame_4k_vs_oracle(ctx):
    """BASELINE.json configs[2] size: one 3840x2160 frame against the oracle (wide kernels, fused levels, 8-level pyramid)."""
    w, h, n = 3840, 2160, 200
    c1, c2 = synth.gen_pair(w, h)
    g = synth.unit_field(w, h, 16)
    p1, p2 = synth.point_pairs(w, h, n, seed=13, dup=0, oob=0)
    want, _ = O.morph_images(c1, c2, g, p1, p2, 0.4, 0.4, 64)
    got = ctx.morph_images(c1, c2, g, p1, p2, 0.4, 0.4)
    _same("4K frame", got[0] if isinstance(got, tuple) else got, want)


def test_phase_frames_in_flight_and_downloads_match_single_frames():
    """Phase-mode frames run on their own streams, several in flight, and the writer receives them through the pinned
    ring in order: every delivered frame must equal the frame rendered on its own."""
    w, h, n, N = 320, 200, 40, 10
    c1 = synth.textured_bgr(w, h, 81); c2 = synth.textured_bgr(w, h, 82)
    g = synth.unit_field(w, h, 17)
    p1, p2 = synth.point_pairs(w, h, n, seed=11, dup=0, oob=0)
    c = capi.Context(0, number_of_frames=N)
    c.pair_load(c1, c2, g, p1, p2)
    shapes = np.array([j / float(N) for j in range(N)])
    got = []
    c.render_many(shapes, chain=False, write=lambda f: got.append(f.copy()))
    assert len(got) == N
    for j in range(N):
        c.reset()
        one = c.render(float(shapes[j]), float(shapes[j]), chain=False)
        _same(f"phase frame {j}", got[j], one)
    c.close()


@pytest.mark.parametrize("chain", [False, True])
def test_ragged_frames_through_the_download_ring(chain):
    """A frame size whose byte count is no multiple of the ring's slot alignment (317 x 211 x 3 = 200 661 bytes): the ring slots
    are padded, the writer must still see exactly the frame — in phase mode against singly rendered frames, in chained mode
    against the same chain walked one frame at a time."""
    w, h, n, N = 317, 211, 36, 7
    c1 = synth.textured_bgr(w, h, 71); c2 = synth.textured_bgr(w, h, 72)
    g = synth.unit_field(w, h, 19)
    p1, p2 = synth.point_pairs(w, h, n, seed=17, dup=0, oob=0)
    c = capi.Context(0, number_of_frames=N)
    c.pair_load(c1, c2, g, p1, p2)
    shapes = (np.array([capi.lib().poppy_frame_ratio(j, N, -1.0) for j in range(N)]) if chain
              else np.array([(j + 1) / float(N + 1) for j in range(N)]))
    got = []
    c.render_many(shapes, chain=chain, write=lambda f: got.append(f.copy()))
    assert len(got) == N and got[0].shape == (h, w, 3)
    c.reset()
    for j in range(N):
        if not chain:
            c.reset()
        one = c.render(float(shapes[j]), float(shapes[j]), chain=chain)
        _same(f"{'chained' if chain else 'phase'} frame {j}", got[j], one)
    c.close()


def test_two_pairs_concurrently_match_sequential():
    """Independent contexts driven from two host threads (the batched configuration) give the frames of a lone run."""
    import threading
    w, h, n, N = 256, 192, 30, 8
    pairs = []
    for k in range(2):
        c1 = synth.textured_bgr(w, h, 91 + k); c2 = synth.textured_bgr(w, h, 95 + k)
        pairs.append((c1, c2, synth.unit_field(w, h, 18 + k), *synth.point_pairs(w, h, n, seed=21 + k, dup=0, oob=0)))
    ratios = np.array([capi.lib().poppy_frame_ratio(j, N, -1.0) for j in range(N)])

    def run(pair, out):
        c = capi.Context(0, number_of_frames=N)
        c.pair_load(*pair)
        c.render_many(ratios, chain=True, write=lambda f: out.append(f.copy()))
        c.close()
    alone = [[], []]
    for k in range(2):
        run(pairs[k], alone[k])
    both = [[], []]
    th = [threading.Thread(target=run, args=(pairs[k], both[k])) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        assert len(both[k]) == N
        for j in range(N):
            _same(f"pair {k} frame {j}", both[k][j], alone[k][j])


def test_warp_kernel_selection(ctx):
    """The tiled warp kernel takes frames whose width is a multiple of 4; other widths use the general kernel.
    Both are checked bit for bit against the oracle by the tests above; this one pins which ran."""
    plain = capi.Context(0)                       # outside debug mode the raster is fused into the warp kernel (kind 2)
    for (w, h), kind in (((640, 360), 1), ((97, 61), 0)):
        c1 = synth.textured_bgr(w, h, 5); c2 = synth.textured_bgr(w, h, 6)
        g = synth.unit_field(w, h, 3)
        p1, p2 = synth.point_pairs(w, h, 12, seed=1, dup=0, oob=0)
        a, _ = ctx.morph_images(c1, c2, g, p1, p2, 0.5, 0.5)
        assert ctx.last_warp_kind() == (kind and TILED)
        b, _ = plain.morph_images(c1, c2, g, p1, p2, 0.5, 0.5)
        assert plain.last_warp_kind() == (2 if kind and TILED and os.environ.get("POPPY_HIP_IDMAP") is None else (kind and TILED))
        _same("fused vs id-map path", b, a)
    plain.close()


@pytest.mark.parametrize("w,h", [(640, 480), (1000, 96), (1920, 1080)])
def test_strong_deformation_vs_oracle(ctx, w, h):
    """Second point set rotated 35 degrees and scaled 0.8 about the centre: steep per-triangle matrices, footprints far
    from their pixels, many triangles per tile (record-cache slot collisions), large out-of-image regions."""
    rng = np.random.default_rng(w * 7 + h)
    n = 300 if w * h > 500000 else 120
    p1 = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
    a = np.deg2rad(35.0)
    c = np.array([(w - 1) / 2, (h - 1) / 2], np.float32)
    R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]], np.float32) * np.float32(0.8)
    p2 = ((p1 - c) @ R.T + c + rng.normal(0, 2.0, (n, 2))).astype(np.float32)
    p2[:, 0] = np.clip(p2[:, 0], 0, w - 1); p2[:, 1] = np.clip(p2[:, 1], 0, h - 1)
    corners = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]], np.float32)
    p1 = np.concatenate([p1, corners]); p2 = np.concatenate([p2, corners])
    c1 = synth.textured_bgr(w, h, 21); c2 = synth.textured_bgr(w, h, 22)
    g = synth.unit_field(w, h, 9)
    for s in (0.25, 0.8):
        want, wmp, d = O.morph_images(c1, c2, g, p1, p2, s, s, 64, debug=True)
        got, gmp = ctx.morph_images(c1, c2, g, p1, p2, s, s)
        assert ctx.last_warp_kind() == TILED
        for name in ("triMap", "trImg1", "trImg2"):
            _same(f"{name} {w}x{h} s={s}", ctx.fetch(name), d[name])
        _same("frame", got, want)


@pytest.mark.parametrize("w,h", [(160, 96), (97, 61)])
def test_id_map_is_not_cleared_between_frames(w, h):
    """Outside debug mode the triangle-id map is never cleared: every frame writes its ids above a growing frame tag and
    reads older values as "no triangle" (poppy_amd/csrc/kernels.h: launch_raster).  The point sets leave a wide
    uncovered margin (id 0 = identity map there) and the triangles move from frame to frame, so stale ids would show;
    with two frame slots, 4300 frames take each slot across the tag wrap-around (2047 frames).  Checked against debug-mode
    frames (plain map, cleared by memset), which the tests above pin to the oracle.  160x96 runs the tiled kernel, 97x61 the general one."""
    rng = np.random.default_rng(w)
    n = 14
    p1 = np.stack([rng.uniform(0.3 * w, 0.7 * w, n), rng.uniform(0.3 * h, 0.7 * h, n)], 1).astype(np.float32)
    p2 = (p1 + rng.normal(0, 4.0, (n, 2))).astype(np.float32)
    c1 = synth.textured_bgr(w, h, 31); c2 = synth.textured_bgr(w, h, 32)
    g = synth.unit_field(w, h, 4)
    shapes = [0.15, 0.5, 0.85, 0.3]
    ref = capi.Context(0); ref.set_debug(True)
    ref.pair_load(c1, c2, g, p1, p2)
    want = [ref.render(s, s, chain=False) for s in shapes]
    assert (ref.fetch("triMap") == 0).mean() > 0.3           # the uncovered margin is really there
    ref.close()
    os.environ["POPPY_HIP_SLOTS"] = "2"
    try:
        c = capi.Context(0)
    finally:
        del os.environ["POPPY_HIP_SLOTS"]
    c.pair_load(c1, c2, g, p1, p2)
    check = set(range(0, 6)) | set(range(4084, 4108)) | {4299}
    for j in range(4300):
        k = (j * 7) % len(shapes) if j % 3 else j % len(shapes)
        got = c.render(shapes[k], shapes[k], chain=False, fetch=j in check)
        if j in check:
            _same(f"{w}x{h} frame {j}", got, want[k])
    c.close()


def test_unsharp_kernel_forms_on_every_geometry():
    """launch_unsharp picks the streaming kernel (kernels_unsharp_stream.hip) from 4 Mpx up and the tile kernel below; the switches
    POPPY_UNSHARP_STREAM / POPPY_UNSHARP_TILE force one of them (read once per process).  The fixture, 1080p, ragged-size and chained
    tests of this file once more with the streaming kernel forced, and the 4K frame of the reference with the tile kernel forced."""
    import subprocess
    import sys
    if os.environ.get("POPPY_UNSHARP_STREAM") or os.environ.get("POPPY_UNSHARP_TILE"):
        pytest.skip("a kernel form is already forced in this process")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_bstage.py"), "-q", "-x", "-m", "gpu", "-k",
                        "fixtures or 1080p or ragged or chained_sequence"], env=dict(os.environ, POPPY_UNSHARP_STREAM="1"),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_sequences.py"), "-q", "-x", "-m", "gpu", "-k", "cfg3_4k"],
                       env=dict(os.environ, POPPY_UNSHARP_TILE="1"), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_pyramid_launch_forms_stay_exact():
    """The small pyramid levels' launches come in three forms with the same bits: the way up as ONE cone launch (k_collapse_cone, the default since round 6),
    as round 5's pairs of levels (POPPY_HIP_NOCONE: k_collapse2 + k_collapse_level), and one launch per level (POPPY_HIP_NOFUSE: also no k_pyrdown2).  The
    fixture, ragged-size, 1080p and chained tests of this file once more under each switch (read once per process)."""
    import subprocess
    import sys
    if os.environ.get("POPPY_HIP_NOCONE") or os.environ.get("POPPY_HIP_NOFUSE"):
        pytest.skip("a form is already forced in this process")
    here = os.path.dirname(os.path.abspath(__file__))
    for form in ({"POPPY_HIP_NOCONE": "1"}, {"POPPY_HIP_NOFUSE": "1"}):
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_bstage.py"), "-q", "-x", "-m", "gpu", "-k",
                            "fixtures or 1080p or ragged or chained_sequence"], env=dict(os.environ, **form), capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, str(form) + "\n" + r.stdout[-3000:] + r.stderr[-1000:]
