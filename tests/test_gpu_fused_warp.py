"""The fused raster + map + remap kernel (k_warp_bin: triangle ids rasterised per tile in LDS, no id map in HBM) — the path
every frame takes outside debug mode — against the reference fixtures and the oracle: both warped sources, the blend mask,
the Laplacian blend and the frame, bit for bit.  (tests/test_gpu_bstage.py runs in debug mode, i.e. on the id-map path.)"""
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from poppy_amd import capi, synth

pytestmark = pytest.mark.gpu
FUSED = 2 if os.environ.get("POPPY_HIP_IDMAP") is None and os.environ.get("POPPY_HIP_GENERALWARP") is None else None


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def _same(name, got, want):
    assert got.shape == want.shape, name
    neq = _bits(got) != _bits(want)
    if neq.any():
        idx = np.argwhere(neq)
        raise AssertionError(f"{name}: {len(idx)} of {got.size} elements differ, first at {idx[0]}")


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", ["b_64x48", "b_256x256", "b_509x381", "b_1920x1080"])
def test_fixture_stages(ctx, case):
    inp = G.bstage_inputs(case)
    w, h, n, ratios, levels = G.make_inputs.BSTAGE[case]
    for k, (s, m) in enumerate(ratios):
        out, mp = ctx.morph_images(inp["c1"], inp["c2"], inp["gabor2"], inp["pts1"], inp["pts2"], s, m)
        if FUSED and w % 4 == 0:
            assert ctx.last_warp_kind() == FUSED
        G.check(case, f"f{k}_morphedPoints", mp)
        for name in ("trImg1", "trImg2", "lbmask", "lapBlend"):
            G.check(case, f"f{k}_{name}", ctx.fetch(name), what=f"ratio {s}")
        G.check(case, f"f{k}_frame", out)


@pytest.mark.parametrize("w,h,n", [(640, 480, 120), (1000, 96, 120), (1920, 1080, 300), (3840, 2160, 500)])
def test_strong_deformation_vs_oracle(ctx, w, h, n):
    """Second point set rotated 35 degrees and scaled 0.8: steep matrices, thin triangles crossing many tiles, large regions
    whose footprints leave the image (the byte-wise border path).  3840x2160 runs the 128 x 8 tile."""
    rng = np.random.default_rng(w * 7 + h)
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
    for s in ((0.25, 0.8) if w < 3000 else (0.4,)):
        want, wmp, d = O.morph_images(c1, c2, g, p1, p2, s, s, 64, debug=True)
        got, gmp = ctx.morph_images(c1, c2, g, p1, p2, s, s)
        if FUSED:
            assert ctx.last_warp_kind() == FUSED
        for name in ("trImg1", "trImg2"):
            _same(f"{name} {w}x{h} s={s}", ctx.fetch(name), d[name])
        _same("frame", got, want)


@pytest.mark.parametrize("w,h,n", [(256, 192, 1500), (512, 64, 900)])
def test_many_triangles_per_tile(ctx, w, h, n):
    """Thousands of triangles on a small image: every tile's list is longer than one pass of the kernel (16 triangles), so the
    multi-pass path with global indices runs; many triangles are only an outline (zero area after truncation)."""
    rng = np.random.default_rng(n)
    p1 = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
    p2 = (p1 + rng.normal(0, 1.5, (n, 2))).astype(np.float32)
    p2[:, 0] = np.clip(p2[:, 0], 0, w - 1); p2[:, 1] = np.clip(p2[:, 1], 0, h - 1)
    corners = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]], np.float32)
    p1 = np.concatenate([p1, corners]); p2 = np.concatenate([p2, corners])
    c1 = synth.textured_bgr(w, h, 3); c2 = synth.textured_bgr(w, h, 4)
    g = synth.unit_field(w, h, 5)
    want, wmp, d = O.morph_images(c1, c2, g, p1, p2, 0.5, 0.5, 64, debug=True)
    got, gmp = ctx.morph_images(c1, c2, g, p1, p2, 0.5, 0.5)
    if FUSED:
        assert ctx.last_warp_kind() == FUSED
    _same("trImg1", ctx.fetch("trImg1"), d["trImg1"])
    _same("trImg2", ctx.fetch("trImg2"), d["trImg2"])
    _same("frame", got, want)


def test_uncovered_margin_and_few_triangles(ctx):
    """Point sets that leave a wide margin uncovered (identity map there: local index 0) and tiles without any triangle."""
    w, h, n = 320, 200, 9
    rng = np.random.default_rng(5)
    p1 = np.stack([rng.uniform(0.35 * w, 0.65 * w, n), rng.uniform(0.35 * h, 0.65 * h, n)], 1).astype(np.float32)
    p2 = (p1 + rng.normal(0, 5.0, (n, 2))).astype(np.float32)
    c1 = synth.textured_bgr(w, h, 31); c2 = synth.textured_bgr(w, h, 32)
    g = synth.unit_field(w, h, 4)
    for s in (0.2, 0.7):
        want, wmp, d = O.morph_images(c1, c2, g, p1, p2, s, s, 64, debug=True)
        got, _ = ctx.morph_images(c1, c2, g, p1, p2, s, s)
        assert (d["triMap"] == 0).mean() > 0.5
        _same("trImg1", ctx.fetch("trImg1"), d["trImg1"])
        _same("trImg2", ctx.fetch("trImg2"), d["trImg2"])
        _same("frame", got, want)
