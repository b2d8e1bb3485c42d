"""GPU parity of the once-per-pair stage: ORB detect kernels + host selection vs reference fixtures / oracle."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from poppy_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", ["o_256x256", "o_640x480", "o_1920x1080"])
def test_orb_detect_vs_reference(ctx, case):
    inp = G.orb_inputs(case)
    w, h, nfs = G.make_inputs.ORB[case]
    for nf in nfs:
        for im in ("1", "2"):
            G.check(case, f"n{nf}_kp{im}", ctx.orb_detect(inp["g" + im], nf))


@pytest.mark.parametrize("w,h,nf", [(97, 80, 50), (333, 211, 200), (1000, 70, 300), (64, 64, 100)])
def test_orb_detect_vs_oracle_ragged(ctx, w, h, nf):
    g = synth.textured_gray(w, h, 3)
    want = O.orb_detect(g, nf)
    got = ctx.orb_detect(g, nf)
    assert got.shape == want.shape and (got.view(np.uint32) == want.view(np.uint32)).all()


def test_orb_on_reference_prefiltered_images(ctx):
    case = "a_512x384_chain"
    nf = int(G.full(case, "detail")[3])
    G.check(case, "kp1", ctx.orb_detect(G.full(case, "g1"), nf))
    G.check(case, "kp2", ctx.orb_detect(G.full(case, "g2"), nf))


def test_pair_begin_prefiltered_reproduces_reference_frames(ctx):
    """From the reference's ORB inputs + gabor2 (fixtures) to the first chained frames of poppy::morph."""
    case = "a_256x256_chain"
    inp = G.astage_inputs(case)
    nf = int(G.full(case, "detail")[3])
    gabor2 = G.full(case, "gabor2")
    c = capi.Context(0, number_of_frames=int(inp["cfg"][0]))
    c.pair_begin_prefiltered(inp["img1"], inp["img2"], G.full(case, "g1"), G.full(case, "g2"), gabor2, nf)
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1)
    G.check(case, "prepared2", p2)
    frames = c.morph_frames(-1.0)
    assert len(frames) == int(inp["cfg"][0])
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
    c.close()


@pytest.mark.parametrize("case,nf", [("o_256x256", 300), ("o_256x256", 516), ("o_640x480", 500), ("o_1920x1080", 516)])
def test_orb_describe_and_hamming_vs_opencv(ctx, case, nf):
    """ORB::compute + BFMatcher(NORM_HAMMING).match (no call site in Poppy; pinned against OpenCV's outputs)."""
    inp = G.orb_inputs(case)
    descs = {}
    for im in ("1", "2"):
        kp = G.full(case, f"n{nf}_kp{im}")
        d = ctx.orb_describe(inp["g" + im], kp)
        G.check(case, f"n{nf}_desc{im}", d)
        descs[im] = d
    G.check(case, f"n{nf}_bfmatch", ctx.hamming_match(descs["1"], descs["2"]))
    # the reference's descriptor-matching sketch (src/experiments.hpp:14-144) on the same descriptors
    from poppy_amd import capi
    k12, k21 = ctx.hamming_knn2(descs["1"], descs["2"]), ctx.hamming_knn2(descs["2"], descs["1"])
    G.check(case, f"n{nf}_knn12", k12)
    G.check(case, f"n{nf}_knn21", k21)
    G.check(case, f"n{nf}_sym", capi.ratio_symmetry(k12, k21, 0.7))


def test_knn2_ties_and_ragged(ctx):
    rng = np.random.RandomState(6)
    q = rng.randint(0, 256, (131, 32)).astype(np.uint8)
    t = rng.randint(0, 256, (1537, 32)).astype(np.uint8)
    t[900] = t[3]; t[17] = q[5]; t[1200] = q[5]; t[1536] = q[5]      # duplicates: ordered by train index
    got = ctx.hamming_knn2(q, t)
    assert np.array_equal(got, O.hamming_knn2(q, t))
    assert got[5].tolist() == [17, 0, 1200, 0]
    assert np.array_equal(ctx.hamming_knn2(q, t[:1]), O.hamming_knn2(q, t[:1]))
    assert (ctx.hamming_knn2(q, t[:0]) == -1).all()
    low = (rng.randint(0, 4, (300, 32)) == 0).astype(np.uint8)          # few distinct distances: many ties
    assert np.array_equal(ctx.hamming_knn2(low[:100], low), O.hamming_knn2(low[:100], low))


def test_hamming_ties_and_ragged(ctx):
    rng = np.random.RandomState(5)
    q = rng.randint(0, 256, (77, 32)).astype(np.uint8)
    t = rng.randint(0, 256, (1301, 32)).astype(np.uint8)
    t[900] = t[3]; t[17] = q[5]; t[1200] = q[5]          # exact duplicates: the lowest train index must win
    got = ctx.hamming_match(q, t)
    want = O.hamming_match(q, t)
    assert (got == want).all()
    assert got[5, 1] == 17 and got[5, 2] == 0
    assert len(ctx.hamming_match(q, t[:0])) == 0
