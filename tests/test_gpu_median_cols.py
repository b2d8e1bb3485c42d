"""GPU parity of the column-histogram median (kernels_median_cols.hip; cv::medianBlur as Extractor::foreground calls it, src/extractor.cpp:149,
OCV/imgproc/src/median_blur.simd.hpp:84-346) through poppy_hip_median_blur: every form of the kernel against the oracle's exact median, on noise (every
tile on all 256 values), on few-valued content (tiles on ranks), on mixtures of both within one image, at ragged sizes and for every window of the chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from poppy_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _noise(w, h, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


def _few(w, h, seed, values):
    """blocky content with a handful of far-apart values: the rank form's case"""
    rng = np.random.default_rng(seed)
    vals = rng.choice(256, values, replace=False).astype(np.uint8)
    coarse = rng.integers(0, values, ((h + 6) // 7, (w + 10) // 11))
    img = vals[np.kron(coarse, np.ones((7, 11), np.int64))[:h, :w]]
    speck = rng.random((h, w)) < 0.03                     # a few single pixels of other values of the same set
    img[speck] = vals[rng.integers(0, values, int(speck.sum()))]
    return np.ascontiguousarray(img)


def _mixed(w, h, seed):
    """left part few-valued, right part noise, a ramp between: tiles of both kinds and tiles whose footprint straddles them"""
    img = _few(w, h, seed, 5)
    img[:, w // 2:] = _noise(w, h, seed + 1)[:, w // 2:]
    ramp = (np.arange(w) * 255 // max(w - 1, 1)).astype(np.uint8)
    img[h // 3: h // 3 + max(h // 6, 1), :] = ramp
    return img


def _wide(w, h, seed):
    """every tile holds all 256 values AND medians from one end of them to the other (a steep ramp under noise): no window of 128 ranks holds a
    tile's medians, the windows below and above the first are needed"""
    rng = np.random.default_rng(seed)
    ramp = (np.arange(w) % 90) * 255 // 89
    img = np.clip(ramp[None, :] + rng.integers(-40, 41, (h, w)), 0, 255).astype(np.uint8)
    img[::7, ::5] = rng.integers(0, 256, img[::7, ::5].shape, dtype=np.uint8)
    return img


FORMS = (1, 2, 3, 4, 5, 6, 7)


@pytest.mark.parametrize("ksize", [3, 9, 17, 25, 33, 41, 49, 57, 65, 73, 81, 89])
def test_every_window_of_the_chain_on_three_kinds_of_content(ctx, ksize):
    import oracle_lib as O
    for img in (_noise(333, 129, ksize), _few(333, 129, ksize, 7), _mixed(333, 129, ksize), _wide(333, 129, ksize)):
        want = O.median_blur_u8(img, ksize)
        for form in FORMS:
            got = ctx.median_blur(img, ksize, form)
            assert np.array_equal(got, want), f"ksize {ksize} form {form}: {np.count_nonzero(got != want)} pixels differ"


@pytest.mark.parametrize("w,h", [(1, 1), (5, 3), (64, 48), (97, 61), (130, 23), (33, 150), (145, 40), (289, 35), (640, 37), (511, 90)])
def test_ragged_sizes(ctx, w, h):
    import oracle_lib as O
    for ksize in (5, 33, 89):
        for img in (_noise(w, h, w + h), _few(w, h, w * h, 3), _mixed(w, h, w), _wide(w, h, h)):
            want = O.median_blur_u8(img, ksize)
            for form in FORMS:
                got = ctx.median_blur(img, ksize, form)
                assert np.array_equal(got, want), f"{w}x{h} ksize {ksize} form {form}: {np.count_nonzero(got != want)} pixels differ"


def test_exactly_64_and_65_values_per_tile(ctx):
    """the rank form takes a tile whose footprint holds at most 64 different values: both sides of that limit, and one value only"""
    import oracle_lib as O
    rng = np.random.default_rng(64)
    for values in (1, 2, 63, 64, 65, 66, 127, 128, 129, 130, 200, 255, 256):
        vals = np.sort(rng.choice(256, values, replace=False)).astype(np.uint8)
        img = vals[rng.integers(0, values, (150, 300))]
        for ksize in (9, 41, 89):
            want = O.median_blur_u8(img, ksize)
            for form in (2, 3, 4, 5, 6, 7):
                assert np.array_equal(ctx.median_blur(img, ksize, form), want), f"{values} values, ksize {ksize}, form {form}"


def test_1080p_chain_of_medians_on_photograph_and_shapes(ctx):
    """the progressive chain of Extractor::foreground (median 9 of the grey image, 17 of that, ...) at 1080p: every link, forms 2 and 1 agree.
    A CONSISTENCY check between the library's two kernels (the oracle's median takes minutes at this size), not parity evidence by itself: both forms
    are compared with the oracle at the sizes it finishes in seconds by the tests beside this one, and form 1 at 1080p by test_gpu_prefilter.py."""
    from poppy_amd import synth
    for name, bgr in (("shapes", synth.gen(1920, 1080, 1234)), ("photo", synth.photo_pair(1920, 1080)[0])):
        cur = np.ascontiguousarray(bgr[:, :, 1])
        for i in range(1, 12):
            a = ctx.median_blur(cur, 8 * i + 1, 2)
            b = ctx.median_blur(cur, 8 * i + 1, 1)
            assert np.array_equal(a, b), f"{name}: link {i} (ksize {8 * i + 1}): {np.count_nonzero(a != b)} pixels differ"
            cur = a


def test_foreground_chain_takes_the_same_bytes_with_either_median(ctx):
    """Extractor::foreground end to end with the medians forced to one form or the other (POPPY_MED_COLS_MIN is read once per process, so the
    two settings are compared through the stage planes of a run against the oracle in test_gpu_prefilter.py; here: the chain's default against form 1)"""
    import oracle_lib as O
    img = _mixed(420, 200, 3)
    for ksize in (9, 49, 89):
        assert np.array_equal(ctx.median_blur(img, ksize, 0), O.median_blur_u8(img, ksize))
