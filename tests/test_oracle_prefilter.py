"""Oracle of the pre-ORB filter chain, part 1 (Extractor::foreground, src/extractor.cpp:136-229) against the outputs
of the reference itself (tests/golden/f_*.npz: every intermediate of three images, captured by oracle/golden_gen)."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

CASES = sorted(G.make_inputs.FSTAGE)


@pytest.mark.parametrize("case", CASES)
def test_foreground_every_stage(case):
    img = G.make_inputs.fstage_inputs(case)["img1"]
    got = O.foreground(img)
    names = list(G.entries(case))
    assert set(names) == set(got), sorted(set(names) ^ set(got))
    # in pipeline order, so that the first mismatch names the stage that broke
    order = ["grey", "flow0", "acc0"] + [f"{s}{i}" for i in range(1, 13) for s in ("med", "flow", "acc", "blur")] + \
            ["log20", "lin", "logged", "finalMask", "masked", "foreground"]
    for name in order:
        G.check(case, name, got[name])


def test_primitives_from_reference_intermediates():
    """Each primitive on the reference's OWN input for that stage (so an upstream error cannot hide a downstream one)."""
    case = "f_160x120"
    for i in range(1, 13):
        last = G.full(case, "grey") if i == 1 else G.full(case, f"med{i - 1}")
        assert np.array_equal(O.median_blur_u8(last, (i - 1) * 8 + 1), G.full(case, f"med{i}")), f"median k={(i - 1) * 8 + 1}"
        assert np.array_equal(O.gaussian_blur23_u8(G.full(case, f"acc{i}")), G.full(case, f"blur{i}")), f"gaussian {i}"
    assert np.array_equal(O.equalize_hist(G.full(case, "masked")), G.full(case, "foreground"))
    lin, logged = G.full(case, "lin"), G.full(case, "logged")
    idx = np.linspace(0, lin.size - 1, 4000).astype(int)
    for v, want in zip(lin.ravel()[idx], logged.ravel()[idx]):
        assert np.float32(O.log32f(v)).tobytes() == np.float32(want).tobytes()


def test_median_is_the_true_median_on_ragged_sizes():
    rng = np.random.default_rng(3)
    for (h, w, k) in [(5, 7, 9), (1, 13, 17), (9, 1, 9), (12, 10, 25)]:
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        r = k // 2
        pad = np.pad(a, r, mode="edge")
        want = np.array([[np.sort(pad[y:y + k, x:x + k].ravel())[k * k // 2] for x in range(w)] for y in range(h)], np.uint8)
        assert np.array_equal(O.median_blur_u8(a, k), want)
