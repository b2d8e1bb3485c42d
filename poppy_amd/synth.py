"""Integer-defined synthetic inputs for the morph hot path.

Everything here is plain integer arithmetic so that the same bytes can be
produced on any box without OpenCV: the golden generator (container only),
the parity tests and bench.py all draw their inputs from these functions.

gen_pair(w, h)            -> (imgA, imgB)  8UC3 BGR, flat discs/squares on grey
textured_gray(w, h, seed) -> 8UC1 image with block + pixel noise (many FAST corners)
unit_field(w, h, seed)    -> f32x3 field in [0,1] (stand-in for gabor2 in B-stage tests)
point_pairs(w, h, n, seed)-> two float32 point sets (n x 2) with a smooth displacement
"""
import numpy as np

_M64 = (1 << 64) - 1


class XorShift64Star:
    """xorshift64* (Vigna); uniform(lo, hi) = lo + (next() >> 33) % (hi - lo)."""

    def __init__(self, seed):
        self.x = (seed & _M64) or 0x9E3779B97F4A7C15

    def next(self):
        x = self.x
        x ^= x >> 12
        x ^= (x << 25) & _M64
        x ^= x >> 27
        self.x = x
        return (x * 0x2545F4914F6CDD1D) & _M64

    def uniform(self, lo, hi):
        return lo + (self.next() >> 33) % (hi - lo)


def gen(w, h, seed, dx=0, dy=0):
    """40 filled shapes on a (30,30,30) background; odd i = disc, even i = square."""
    img = np.full((h, w, 3), 30, dtype=np.uint8)
    rng = XorShift64Star(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    for i in range(40):
        cx = rng.uniform(w // 8, 7 * w // 8) + dx
        cy = rng.uniform(h // 8, 7 * h // 8) + dy
        r = rng.uniform(max(h // 40, 1), max(h // 10, 2))
        col = [rng.uniform(40, 255) for _ in range(3)]
        if i & 1:
            m = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        else:
            m = (np.abs(xx - cx) <= r) & (np.abs(yy - cy) <= r)
        img[m] = col
    return img


def gen_pair(w, h, seed=1234):
    return gen(w, h, seed, 0, 0), gen(w, h, seed, w // 50, w // 100)


def _hash32(x, y, seed):
    h = (x.astype(np.uint64) * np.uint64(73856093)) ^ (y.astype(np.uint64) * np.uint64(19349663)) \
        ^ np.uint64((seed * 83492791) & 0xFFFFFFFF)
    h &= np.uint64(0xFFFFFFFF)
    h = ((h ^ (h >> np.uint64(13))) * np.uint64(0x5BD1E995)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return h


def textured_gray(w, h, seed=7):
    """Grey version of gen() plus 4x4-block noise (+-40) and pixel noise (+-6)."""
    base = gen(w, h, seed).astype(np.int64)
    g = (base[..., 0] * 29 + base[..., 1] * 150 + base[..., 2] * 77) >> 8
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    nb = (_hash32(xx >> 2, yy >> 2, seed).astype(np.int64) % 81) - 40
    npx = (_hash32(xx, yy, seed + 1).astype(np.int64) % 13) - 6
    return np.clip(g + nb + npx, 0, 255).astype(np.uint8)


def textured_bgr(w, h, seed=7):
    """Colour image with per-channel texture (remap / blend tests)."""
    base = gen(w, h, seed).astype(np.int64)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    out = np.empty((h, w, 3), dtype=np.uint8)
    for c in range(3):
        nb = (_hash32(xx >> 3, yy >> 3, seed + 10 * c).astype(np.int64) % 61) - 30
        npx = (_hash32(xx, yy, seed + 10 * c + 1).astype(np.int64) % 17) - 8
        out[..., c] = np.clip(base[..., c] + nb + npx, 0, 255)
    return out


def unit_field(w, h, seed=11):
    """f32x3 field in [0,1]: k/1024 with k from 16x16-block noise + pixel noise."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    out = np.empty((h, w, 3), dtype=np.float32)
    for c in range(3):
        k = (_hash32(xx >> 4, yy >> 4, seed + c).astype(np.int64) % 769) \
            + (_hash32(xx, yy, seed + 3 + c).astype(np.int64) % 256)
        out[..., c] = (k.astype(np.float64) / 1024.0).astype(np.float32)
    return out


def point_pairs(w, h, n, seed=5, dup=2, oob=2):
    """n correspondences: p1 on quarter-pixel lattice, p2 = p1 + smooth shift.

    `dup` exact duplicates and `oob` slightly out-of-range points are appended to
    exercise clip_points / make_uniq (reference src/util.cpp:453-460, 541-548).
    The 4 image corners are appended last, as Matcher::prepare does.
    """
    rng = XorShift64Star(seed)
    p1 = np.empty((n, 2), dtype=np.float32)
    p2 = np.empty((n, 2), dtype=np.float32)
    for i in range(n):
        x = rng.uniform(4 * 2, 4 * (w - 2)) / 4.0
        y = rng.uniform(4 * 2, 4 * (h - 2)) / 4.0
        sx = (w / 50.0) * (0.5 + y / h) + (rng.uniform(0, 17) - 8) / 4.0
        sy = (w / 100.0) * (0.5 + x / w) + (rng.uniform(0, 17) - 8) / 4.0
        p1[i] = (x, y)
        p2[i] = (min(max(x + sx, 0.0), w - 1.0), min(max(y + sy, 0.0), h - 1.0))
    extra1, extra2 = [], []
    for k in range(dup):
        j = rng.uniform(0, n)
        extra1.append(p1[j].copy())
        extra2.append(p2[j].copy())
    for k in range(oob):
        extra1.append(np.array([w + 3.5, h / 2.0 + k], dtype=np.float32))
        extra2.append(np.array([w / 2.0 + k, -2.25], dtype=np.float32))
    corners = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]], dtype=np.float32)
    a = np.vstack([p1] + [e[None] for e in extra1] + [corners]).astype(np.float32)
    b = np.vstack([p2] + [e[None] for e in extra2] + [corners]).astype(np.float32)
    return a, b


def upscale_bgr(img, w, h):
    """Integer bilinear resize of a u8 image to w x h (16.16 fixed-point source coordinates, pixel centres aligned, result rounded to nearest):
    defined here, in integers, so that every box makes the same pixels from the committed photo fixture (tests/golden/photo_pair_720x405.npz)."""
    sh, sw = img.shape[:2]
    def axis(n_out, n_in):
        c = ((2 * np.arange(n_out, dtype=np.int64) + 1) * n_in * 32768) // n_out - 32768       # source coordinate of the pixel centre, 16.16
        c = np.clip(c, 0, (n_in - 1) << 16)
        i0 = (c >> 16).astype(np.int64)
        return i0, np.minimum(i0 + 1, n_in - 1), (c & 0xffff).astype(np.int64)
    y0, y1, fy = axis(h, sh)
    x0, x1, fx = axis(w, sw)
    a = img.astype(np.int64)
    top = a[y0][:, x0] * (65536 - fx)[None, :, None] + a[y0][:, x1] * fx[None, :, None]
    bot = a[y1][:, x0] * (65536 - fx)[None, :, None] + a[y1][:, x1] * fx[None, :, None]
    out = (top * (65536 - fy)[:, None, None] + bot * fy[:, None, None] + (1 << 31)) >> 32
    return np.ascontiguousarray(out.astype(np.uint8))            # (the fancy indexing above leaves a transposed memory order)


def photo_pair(w, h):
    """The reference's own sample photographs (images/amir1.jpg, amir2.jpg; committed as pixels: tests/golden/photo_pair_720x405.npz) at w x h."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "photo_pair_720x405.npz"))
    return upscale_bgr(z["a"], w, h), upscale_bgr(z["b"], w, h)


def demo_pair(name):
    """One of the reference's own demo pairs of a width that is no multiple of 4 (committed as pixels: tests/golden/demo_pairs.npz, tools/make_demo_fixture.py):
    "numbers" = images/1st.png + 2nd.png (639 x 480), "cars" = images/kindpng1s.png + kindpng2s.png (749 x 480)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demo_pairs.npz"))
    return np.ascontiguousarray(z[name + "_a"]), np.ascontiguousarray(z[name + "_b"])


def fnv1a64(buf):
    """FNV-1a over the raw bytes of an array (used by the golden manifests)."""
    data = np.ascontiguousarray(buf).view(np.uint8).ravel()
    # vectorised FNV is awkward; process in Python ints on chunks via polynomial trick is
    # overkill here, fixtures are small. Use a simple loop over bytes in blocks.
    hsh = 1469598103934665603
    prime = 1099511628211
    for b in data.tobytes():
        hsh = ((hsh ^ b) * prime) & _M64
    return hsh
