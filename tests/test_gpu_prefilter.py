"""GPU parity of poppy_hip_foreground (Extractor::foreground, src/extractor.cpp:136-229) through the C ABI:
every intermediate against the reference's own outputs (tests/golden/f_*), ragged sizes and 1080p against the oracle."""
import os

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

ORDER = ["grey", "flow0", "acc0"] + [f"{s}{i}" for i in range(1, 13) for s in ("med", "flow", "acc", "blur")] + \
        ["lin", "logged", "finalMask", "masked", "foreground"]


@pytest.fixture(scope="module")
def ctx():
    from poppy_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", sorted(G.make_inputs.FSTAGE))
def test_foreground_every_stage_vs_reference(ctx, case):
    img = G.make_inputs.fstage_inputs(case)["img1"]
    got = ctx.foreground(img, debug=True)
    for name in ORDER:                      # pipeline order: the first mismatch names the stage that broke
        G.check(case, name, got[name])
    assert np.array_equal(ctx.foreground(img), got["foreground"])      # the non-debug path is the same computation


# (256 x 64 and wider in steps of 4: k_acc_gauss23_v4, three accumulate-and-blur steps per launch with reflected halos; the rest: k_acc_gauss23)
@pytest.mark.parametrize("w,h,seed", [(64, 48, 1), (97, 61, 2), (130, 23, 3), (33, 150, 4), (256, 64, 5), (300, 70, 6), (388, 101, 7), (260, 65, 8), (258, 80, 9)])
def test_foreground_ragged_vs_oracle(ctx, w, h, seed):
    import oracle_lib as O
    from poppy_amd import synth
    img = synth.textured_bgr(w, h, seed)
    want = O.foreground(img)
    got = ctx.foreground(img, debug=True)
    for name in ORDER:
        a, b = got[name], want[name]
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        assert same.all(), f"{w}x{h} {name}: {np.count_nonzero(~same)} elements differ"
    # the non-debug path accumulates and blurs the mask in fused launches (k_acc_gauss23 / k_acc_gauss23_v4): same image at ragged sizes too
    assert np.array_equal(ctx.foreground(img), want["foreground"])


def test_foreground_1080p_vs_oracle_final(ctx):
    import oracle_lib as O
    from poppy_amd import synth
    img = synth.gen(1920, 1080, 1234)
    got = ctx.foreground(img)
    want = O.foreground(img)["foreground"]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h,seed", [(160, 120, 5), (97, 61, 6)])
def test_foreground_with_radial_mask_vs_oracle(w, h, seed):
    """Settings::enable_radial_mask (src/extractor.cpp:178-197): the foreground mask times draw_radial_gradiant's mask before the
    log curve.  Every stage against the oracle's restatement (no reference-run fixture exists for this option: DESIGN.md section 2)."""
    import oracle_lib as O
    from poppy_amd import capi, synth
    img = synth.textured_bgr(w, h, seed)
    plain = O.foreground(img)
    O.set_radial_mask(True)
    try:
        want = O.foreground(img)
    finally:
        O.set_radial_mask(False)
    assert not np.array_equal(want["foreground"], plain["foreground"])          # the option does something
    c = capi.Context(0, enable_radial_mask=1)
    got = c.foreground(img, debug=True)
    for name in ORDER:
        a, b = got[name], want[name]
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        assert same.all(), f"{w}x{h} {name}: {np.count_nonzero(~same)} elements differ"
    assert np.array_equal(c.foreground(img), want["foreground"])
    c.close()
    c0 = capi.Context(0)                                                        # and off by default
    assert np.array_equal(c0.foreground(img), plain["foreground"])
    c0.close()


def test_pair_begin_with_radial_mask_vs_oracle():
    """The option end to end: nfeatures and the prepared point pairs of a raw pair with enable_radial_mask, against the oracle's set-up."""
    import oracle_lib as O
    from poppy_amd import capi, synth
    a, b = synth.gen_pair(200, 152, seed=77)
    O.set_radial_mask(True)
    try:
        want = O.pair_setup(a, b)
    finally:
        O.set_radial_mask(False)
    c = capi.Context(0, enable_radial_mask=1)
    nf, det = c.pair_begin(a, b)
    assert nf == want["nfeatures"] and det == want["detail"]
    p1, p2 = c.pair_points()
    assert np.array_equal(p1, want["points1"]) and np.array_equal(p2, want["points2"])
    c.close()


def test_setup_alternative_forms_stay_exact():
    """Switches of the pair set-up that are read once per process, each checked in a process of its own against the same fixtures and oracle:
    POPPY_ACC_STEPS (accumulate-and-blur steps per launch of k_acc_gauss23_v4: 1, 2, 4, 6; 0 = the byte-per-thread kernel; the default is 3),
    POPPY_MED_WAVES / POPPY_MED_SETS (waves per histogram set and sets per launch of the medians),
    POPPY_ORB_GUESS=0 (the detector's first copy brings the counts only, every candidate comes with the second copy),
    POPPY_ORB_CAP=50 (candidate lists far too short for any image: the detector re-allocates them for the counted candidates and runs FAST again —
    what noise-like content does to the default lists: round 4, found by timing the set-up on textured images), POPPY_ORB_KPCAP=64 (the
    same for the keypoint buffers behind the first retainBest, whose ties are unbounded),
    POPPY_GABOR2_FIRST (gabor2 beside the first medians instead of behind the second image's)."""
    import subprocess
    import sys
    forms = [{"POPPY_ACC_STEPS": "0"}, {"POPPY_ACC_STEPS": "1"}, {"POPPY_ACC_STEPS": "2"}, {"POPPY_ACC_STEPS": "4"}, {"POPPY_ACC_STEPS": "6"},
             {"POPPY_MED_WAVES": "1"}, {"POPPY_MED_WAVES": "2"}, {"POPPY_MED_WAVES": "8"}, {"POPPY_MED_SETS": "192"},
             {"POPPY_ORB_GUESS": "0"}, {"POPPY_ORB_CAP": "50"}, {"POPPY_ORB_KPCAP": "64"}, {"POPPY_GABOR2_FIRST": "1"}]
    if any(k in os.environ for f in forms for k in f):
        pytest.skip("a form is already forced in this process")
    here = os.path.dirname(os.path.abspath(__file__))
    for f in forms:
        acc = "POPPY_ACC_STEPS" in f or "POPPY_MED_WAVES" in f or "POPPY_MED_SETS" in f
        args = [os.path.join(here, "test_gpu_prefilter.py"), "-k", "every_stage or ragged or 1080p"] if acc else \
               [os.path.join(here, "test_gpu_astage.py"), os.path.join(here, "test_gpu_sequences.py"), "-k", "orb_detect or cfg1 or cfg2 or sharded"]
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu"] + args, env=dict(os.environ, **f), capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, str(f) + "\n" + r.stdout[-3000:] + r.stderr[-1000:]
