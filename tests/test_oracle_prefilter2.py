"""Oracle + host pieces of the rest of the pre-ORB chain against the reference's outputs (tests/golden/a_*):
exact stages bit for bit, the Gabor bank (DFT-based filter2D in the reference) to a stated tolerance."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

CASES = ["a_256x256_chain", "a_512x384_chain"]


@pytest.mark.parametrize("case", CASES)
def test_radial_gradient_host_exact(case):
    from poppy_amd import capi
    ref = G.full(case, "radial")
    got = capi.radial_gradient(ref.shape[1], ref.shape[0])
    G.check(case, "radial", got)


def test_gabor_kernels_exact():
    case = "a_256x256_chain"
    b31 = O.gabor_bank(31, 5, 2)
    b13 = O.gabor_bank(13, 5, 10)
    G.check(case, "gaborK31_0", b31[0]); G.check(case, "gaborK31_5", b31[5]); G.check(case, "gaborK13_3", b13[3])


@pytest.mark.parametrize("case", CASES)
def test_unsharp_sigma2_grey_exact(case):
    G.check(case, "us1", O.orb_unsharp_gray(G.full(case, "goodFeatures1")))


def test_gabor_bank_direct_within_tolerance_of_reference():
    case = "a_256x256_chain"
    us = G.full(case, "us1")[64:128, 64:128]
    full = G.full(case, "us1")
    # direct sums on a 64x64 window of the image (borders excluded by comparing the interior only)
    got = O.gabor_filter_direct(full[32:160, 32:160], 31, O.gabor_bank(31, 5, 2))[32:96, 32:96]
    ref = G.full(case, "gb1")[64:128, 64:128]
    assert us.shape == got.shape
    assert np.abs(got - ref).max() <= 1e-5            # OpenCV's DFT-based filter2D vs a direct sum in double


@pytest.mark.parametrize("case", sorted(G.make_inputs.MARGIN))
def test_blur_margin_oracle_exact(case):
    inp = G.make_inputs.margin_inputs(case)
    G.check(case, "padded", O.blur_margin(inp["img"], int(inp["cfg"][0]), int(inp["cfg"][1])))
