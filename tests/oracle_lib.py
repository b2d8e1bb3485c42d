"""ctypes binding of oracle/liboracle.so for the tests (and bench.py's cpu_baseline leg).

The oracle is test infrastructure: nothing under poppy_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build()
        _lib = C.CDLL(so)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def clip_points(pts, cols, rows):
    p = _f(pts).copy()
    lib().orc_clip_points(_vp(p), len(p), cols, rows)
    return p


def make_uniq(pts):
    p = _f(pts)
    out = np.empty_like(p)
    n = lib().orc_make_uniq(_vp(p), len(p), _vp(out))
    return out[:n].copy()


def morph_points(a, b, s):
    a, b = _f(a), _f(b)
    out = np.empty_like(a)
    lib().orc_morph_points(_vp(a), _vp(b), len(a), C.c_float(s), _vp(out))
    return out


def delaunay(w, h, pts, max_tris=None):
    p = _f(pts)
    max_tris = max_tris or 2 * len(p) + 16
    out = np.empty((max_tris, 6), np.float32)
    n = lib().orc_delaunay(w, h, _vp(p), len(p), _vp(out), max_tris)
    if n < 0:
        raise ValueError("delaunay failed: %d" % n)
    return out[:n].copy()


def triangle_indices(tri6, pts):
    t, p = _f(tri6), _f(pts)
    out = np.empty((len(t), 3), np.int32)
    n = lib().orc_triangle_indices(_vp(t), len(t), _vp(p), len(p), _vp(out))
    return out[:n].copy()


def triangle_int_points(idx3, pts):
    idx = np.ascontiguousarray(idx3, np.int32)
    p = _f(pts)
    out = np.empty((len(idx), 3, 2), np.int32)
    lib().orc_triangle_int_points(_vp(idx), len(idx), _vp(p), len(p), _vp(out))
    return out


def paint_triangles(w, h, tris):
    t = np.ascontiguousarray(tris, np.int32).reshape(-1, 6)
    out = np.empty((h, w), np.int32)
    lib().orc_paint_triangles(w, h, _vp(t), len(t), _vp(out))
    return out


def invert33(m):
    m = _f(m).reshape(-1, 9)
    out = np.empty_like(m)
    for i in range(len(m)):
        lib().orc_invert33(_vp(m[i]), _vp(out[i]))
    return out.reshape(-1, 3, 3)


def homographies(t1, t2, ratio):
    a = np.ascontiguousarray(t1, np.int32).reshape(-1, 6)
    b = np.ascontiguousarray(t2, np.int32).reshape(-1, 6)
    n = len(a)
    H, M1, M2 = (np.empty((n, 3, 3), np.float32) for _ in range(3))
    lib().orc_homographies(_vp(a), _vp(b), n, C.c_float(ratio), _vp(H), _vp(M1), _vp(M2))
    return H, M1, M2


def create_map(tri_map, mats):
    tm = np.ascontiguousarray(tri_map, np.int32)
    m = _f(mats).reshape(-1, 9)
    h, w = tm.shape
    mx, my = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    lib().orc_create_map(_vp(tm), w, h, _vp(m), len(m), _vp(mx), _vp(my))
    return mx, my


def remap(src, mapx, mapy):
    s = np.ascontiguousarray(src, np.uint8)
    sh, sw = s.shape[:2]
    c = s.shape[2] if s.ndim == 3 else 1
    mx, my = _f(mapx), _f(mapy)
    h, w = mx.shape
    out = np.empty((h, w, c) if s.ndim == 3 else (h, w), np.uint8)
    lib().orc_remap(_vp(s), sw, sh, c, _vp(mx), _vp(my), w, h, _vp(out))
    return out


def bilinear_tab():
    out = np.empty((1024, 4), np.int16)
    lib().orc_bilinear_tab(_vp(out))
    return out


def u8_to_f32(a):
    a = np.ascontiguousarray(a, np.uint8)
    out = np.empty(a.shape, np.float32)
    lib().orc_u8_to_f32(_vp(a), a.size, _vp(out))
    return out


def f32_to_u8(a):
    a = _f(a)
    out = np.empty(a.shape, np.uint8)
    lib().orc_f32_to_u8(_vp(a), a.size, _vp(out))
    return out


def blend_mask(gabor2, mask_ratio):
    g = _f(gabor2)
    h, w = g.shape[:2]
    out = np.empty((h, w), np.float32)
    lib().orc_blend_mask(_vp(g), w, h, C.c_double(mask_ratio), _vp(out))
    return out


def _shape3(a):
    h, w = a.shape[:2]
    c = a.shape[2] if a.ndim == 3 else 1
    return w, h, c


def pyr_down(a):
    a = _f(a)
    w, h, c = _shape3(a)
    out = np.empty(((h + 1) // 2, (w + 1) // 2, c) if a.ndim == 3 else ((h + 1) // 2, (w + 1) // 2), np.float32)
    lib().orc_pyr_down(_vp(a), w, h, c, _vp(out))
    return out


def pyr_up(a, dw, dh):
    a = _f(a)
    w, h, c = _shape3(a)
    out = np.empty((dh, dw, c) if a.ndim == 3 else (dh, dw), np.float32)
    lib().orc_pyr_up(_vp(a), w, h, c, dw, dh, _vp(out))
    return out


def laplacian_blend(l, r, mask, levels):
    l, r, m = _f(l), _f(r), _f(mask)
    h, w = m.shape
    out = np.empty((h, w, 3), np.float32)
    lib().orc_laplacian_blend(_vp(l), _vp(r), _vp(m), w, h, levels, _vp(out))
    return out


def unsharp(a, amount, threshold):
    a = _f(a)
    h, w = a.shape[:2]
    out, blur, med = (np.empty((h, w, 3), np.float32) for _ in range(3))
    lib().orc_unsharp(_vp(a), w, h, C.c_float(amount), C.c_float(threshold), _vp(out), _vp(blur), _vp(med))
    return out, blur, med


def morph_images(c1, c2, gabor2, p1, p2, shape, mask, levels=64, debug=False):
    c1 = np.ascontiguousarray(c1, np.uint8)
    c2 = np.ascontiguousarray(c2, np.uint8)
    g = _f(gabor2)
    p1, p2 = _f(p1), _f(p2)
    h, w = c1.shape[:2]
    n = len(p1)
    out = np.empty((h, w, 3), np.uint8)
    mp = np.empty((n, 2), np.float32)
    max_tris = 2 * n + 16
    ntri = C.c_int(0)
    d = {}
    if debug:
        d = dict(tri6=np.zeros((max_tris, 6), np.float32), idx3=np.zeros((max_tris, 3), np.int32),
                 H=np.zeros((max_tris, 3, 3), np.float32), M1=np.zeros((max_tris, 3, 3), np.float32),
                 M2=np.zeros((max_tris, 3, 3), np.float32), triMap=np.zeros((h, w), np.int32),
                 mapx1=np.zeros((h, w), np.float32), mapy1=np.zeros((h, w), np.float32),
                 mapx2=np.zeros((h, w), np.float32), mapy2=np.zeros((h, w), np.float32),
                 trImg1=np.zeros((h, w, 3), np.uint8), trImg2=np.zeros((h, w, 3), np.uint8),
                 lbmask=np.zeros((h, w), np.float32), lapBlend=np.zeros((h, w, 3), np.float32),
                 unsharp=np.zeros((h, w, 3), np.float32))
    g_ = lambda k: _vp(d[k]) if debug else None
    rc = lib().orc_morph_images(_vp(c1), _vp(c2), _vp(g), w, h, _vp(p1), _vp(p2), n,
                                C.c_double(shape), C.c_double(mask), levels, _vp(out), _vp(mp),
                                max_tris, C.byref(ntri), g_("tri6"), g_("idx3"), g_("H"), g_("M1"), g_("M2"),
                                g_("triMap"), g_("mapx1"), g_("mapy1"), g_("mapx2"), g_("mapy2"),
                                g_("trImg1"), g_("trImg2"), g_("lbmask"), g_("lapBlend"), g_("unsharp"))
    if rc:
        raise ValueError("oracle morph_images failed: %d" % rc)
    if debug:
        nt = ntri.value
        for k in ("idx3", "H", "M1", "M2"):
            d[k] = d[k][:nt]
        d["ntri"] = nt
        return out, mp, d
    return out, mp


def orb_detect(img, nfeatures, with_fast=False):
    g = np.ascontiguousarray(img, np.uint8)
    h, w = g.shape
    kp = np.zeros((nfeatures * 2 + 64, 7), np.float32)
    mf = w * h // 4 + 16
    fast = np.zeros((mf, 3), np.float32)
    nf = C.c_int(0)
    n = lib().orc_orb_detect(_vp(g), w, h, nfeatures, _vp(kp), len(kp), _vp(fast), mf, C.byref(nf))
    if n < 0:
        raise ValueError("orb_detect overflow")
    return (kp[:n].copy(), fast[:nf.value].copy()) if with_fast else kp[:n].copy()


def hamming_match(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros((len(q), 3), np.int32)
    n = lib().orc_hamming_match(_vp(q), len(q), _vp(t), len(t), q.shape[1], _vp(out))
    return out[:n]


def hamming_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros((len(q), 4), np.int32)
    lib().orc_hamming_knn2(_vp(q), len(q), _vp(t), len(t), q.shape[1], _vp(out))
    return out


def ratio_symmetry(knn12, knn21, ratio=0.7):
    """(keep12, keep21, symmetric matches n x 3) of src/experiments.hpp's ratioTest + symmetryTest."""
    knn12 = np.ascontiguousarray(knn12, np.int32); knn21 = np.ascontiguousarray(knn21, np.int32)
    k1 = np.zeros(len(knn12), np.int32); k2 = np.zeros(len(knn21), np.int32)
    out = np.zeros((max(len(knn12), 1), 3), np.int32)
    lib().orc_ratio_symmetry.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    n = lib().orc_ratio_symmetry(_vp(knn12), len(knn12), _vp(knn21), len(knn21), ratio, _vp(k1), _vp(k2), _vp(out))
    return k1, k2, out[:n]


def distance_map(p1, p2):
    p1, p2 = _f(p1), _f(p2)
    out = np.zeros((len(p1), 5), np.float64)
    n = lib().orc_distance_map(_vp(p1), _vp(p2), len(p1), _vp(out))
    return out[:n]


def filter_invalid(p1, p2, cols, rows):
    a, b = _f(p1).copy(), _f(p2).copy()
    n = lib().orc_filter_invalid(_vp(a), _vp(b), len(a), cols, rows)
    return a[:n].copy(), b[:n].copy()


def morph_distance(p1, p2, w, h):
    p1, p2 = _f(p1), _f(p2)
    lib().orc_morph_distance.restype = C.c_double
    return lib().orc_morph_distance(_vp(p1), _vp(p2), len(p1), w, h)


def match_prepare(p1, p2, w, h, tol, imd):
    p1, p2 = _f(p1), _f(p2)
    o1 = np.zeros((len(p1) + 4, 2), np.float32); o2 = np.zeros((len(p1) + 4, 2), np.float32)
    n = lib().orc_match_prepare(_vp(p1), _vp(p2), len(p1), w, h, C.c_double(tol), C.c_double(imd), _vp(o1), _vp(o2))
    return o1[:n].copy(), o2[:n].copy()


def orb_describe(img, kps7, trig_mode=0):
    g = np.ascontiguousarray(img, np.uint8)
    k = _f(kps7)
    h, w = g.shape
    out = np.zeros((len(k), 32), np.uint8)
    lib().orc_orb_describe(_vp(g), w, h, _vp(k), len(k), _vp(out), trig_mode)
    return out


def foreground(bgr):
    """Extractor::foreground for one BGR image: dict with the final image and every intermediate."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    fg = np.zeros((h, w), np.uint8); grey = np.zeros((h, w), np.uint8); masked = np.zeros((h, w), np.uint8)
    stages = np.zeros((50, h, w), np.uint8)
    floats = np.zeros((3, h, w), np.float32)
    ln20 = C.c_float(0)
    lib().orc_foreground(_vp(bgr), w, h, _vp(fg), _vp(grey), _vp(stages), _vp(floats), _vp(masked), C.byref(ln20))
    out = dict(foreground=fg, grey=grey, masked=masked, lin=floats[0], logged=floats[1], finalMask=floats[2],
               log20=np.array([[ln20.value]], np.float32), flow0=stages[0], acc0=stages[1])
    for i in range(12):
        out[f"med{i + 1}"], out[f"flow{i + 1}"], out[f"acc{i + 1}"], out[f"blur{i + 1}"] = stages[2 + 4 * i: 6 + 4 * i]
    return out


def median_blur_u8(a, ksize):
    a = np.ascontiguousarray(a, np.uint8); o = np.zeros_like(a)
    lib().orc_median_blur_u8(_vp(a), a.shape[1], a.shape[0], ksize, _vp(o))
    return o


def gaussian_blur23_u8(a):
    a = np.ascontiguousarray(a, np.uint8); o = np.zeros_like(a)
    lib().orc_gaussian_blur23_u8(_vp(a), a.shape[1], a.shape[0], _vp(o))
    return o


def equalize_hist(a):
    a = np.ascontiguousarray(a, np.uint8); o = np.zeros_like(a)
    lib().orc_equalize_hist(_vp(a), a.shape[1], a.shape[0], _vp(o))
    return o


def log32f(x):
    L = lib(); L.orc_log32f.restype = C.c_float; L.orc_log32f.argtypes = [C.c_float]
    return L.orc_log32f(float(x))


def orb_unsharp_gray(gf):
    gf = np.ascontiguousarray(gf, np.uint8); o = np.zeros(gf.shape, np.float32)
    lib().orc_orb_unsharp_gray(_vp(gf), gf.shape[1], gf.shape[0], _vp(o))
    return o


def gabor_bank(ks, sigma, lambd, gamma=0.04, psi=np.pi / 4):
    o = np.zeros((16, ks, ks), np.float32)
    L = lib(); L.orc_gabor_bank.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
    L.orc_gabor_bank(ks, sigma, lambd, gamma, psi, _vp(o))
    return o


def gabor_filter_direct(src, ks, bank):
    a = _f(src); c = 1 if a.ndim == 2 else a.shape[2]
    o = np.zeros_like(a); b = _f(bank)
    lib().orc_gabor_filter_direct(_vp(a), a.shape[1], a.shape[0], c, ks, _vp(b), _vp(o))
    return o


def blur_margin(img, uw, uh):
    a = np.ascontiguousarray(img, np.uint8); o = np.zeros((uh, uw, 3), np.uint8)
    lib().orc_blur_margin(_vp(a), a.shape[1], a.shape[0], uw, uh, _vp(o))
    return o


# ---- auto-align (oracle/align.cpp) --------------------------------------------------------------------------------
def warp_affine(img, M):
    img = np.ascontiguousarray(img, np.uint8)
    M = np.ascontiguousarray(M, np.float64).reshape(6)
    h, w = img.shape[:2]
    c = img.shape[2] if img.ndim == 3 else 1
    out = np.zeros_like(img)
    lib().orc_warp_affine(_vp(img), w, h, c, _vp(M), _vp(out))
    return out


def rotation_matrix(cx, cy, angle, scale=1.0):
    M = np.zeros(6, np.float64)
    lib().orc_rotation_matrix.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double, C.c_void_p]
    lib().orc_rotation_matrix(cx, cy, angle, scale, _vp(M))
    return M.reshape(2, 3)


def align_prims(x, y, svd_in, tr_m):
    x, y = _f(x), _f(y)
    n = len(x)
    r = dict(mean=np.zeros(2), sumsq=np.zeros(2), gemm=np.zeros((2, 2), np.float32), svd_w=np.zeros((2, 1), np.float32),
             svd_u=np.zeros((2, 2), np.float32), svd_vt=np.zeros((2, 2), np.float32), transform=np.zeros((n, 2), np.float32),
             persp=np.zeros((3, 3)), persp_pts=np.zeros((n, 2), np.float32))
    si = np.ascontiguousarray(svd_in, np.float32); tm = np.ascontiguousarray(tr_m, np.float32)
    lib().orc_align_prims(_vp(x), _vp(y), n, _vp(r["mean"]), _vp(r["sumsq"]), _vp(r["gemm"]), _vp(si), _vp(r["svd_w"]), _vp(r["svd_u"]),
                          _vp(r["svd_vt"]), _vp(tm), _vp(r["transform"]), _vp(r["persp"]), _vp(r["persp_pts"]))
    return r


def procrustes(x, y):
    x, y = _f(x), _f(y)
    rot = np.zeros((2, 2), np.float32); sc = np.zeros(2, np.float32); yp = np.zeros((len(x), 2), np.float32); tr = np.zeros((1, 2), np.float32)
    lib().orc_procrustes(_vp(x), _vp(y), len(x), _vp(rot), _vp(sc), _vp(yp), _vp(tr))
    return dict(rotation=rot, scale=sc[0], error=sc[1], yprime=yp, translation=tr)


def align_step(which, img, p1, p2):
    """which: 'retranslate' | 'reprocrustes' | 'rerotate' | 'auto'.  Returns (image, pts2, distance)."""
    code = {"retranslate": 0, "reprocrustes": 1, "rerotate": 2, "auto": 3}[which]
    im = np.ascontiguousarray(img, np.uint8).copy()
    p1 = _f(p1); q = _f(p2).copy()
    h, w = im.shape[:2]
    lib().orc_align_step.restype = C.c_double
    d = lib().orc_align_step(code, _vp(im), w, h, _vp(p1), _vp(q), len(p1))
    return im, q, d


# ---- round 2: the rest of the pre-ORB chain, the whole pair set-up and poppy::morph in the oracle ---------------------
def dft_detail2(gray):
    g = np.ascontiguousarray(gray, np.uint8)
    L = lib(); L.orc_dft_detail2.restype = C.c_double
    return L.orc_dft_detail2(_vp(g), g.shape[1], g.shape[0])


def radial_gradient(w, h):
    o = np.zeros((h, w), np.float32)
    lib().orc_radial_gradient(w, h, _vp(o))
    return o


def radial_mask(w, h):
    """draw_radial_gradiant as Extractor::foreground uses it under Settings::enable_radial_mask (float, 0..1)."""
    o = np.zeros((h, w), np.float32)
    lib().orc_radial_mask(w, h, _vp(o))
    return o


def set_radial_mask(on):
    """Settings::enable_radial_mask for the oracle's foreground() (and everything built on it); remember to switch it off again."""
    lib().orc_set_radial_mask(int(bool(on)))


def orb_input(good_features):
    gf = np.ascontiguousarray(good_features, np.uint8); o = np.zeros_like(gf)
    lib().orc_orb_input(_vp(gf), gf.shape[1], gf.shape[0], _vp(o))
    return o


def gaussian_taps_fx(n, sigma):
    o = np.zeros(n, np.int32)
    L = lib(); L.orc_gaussian_taps_fx.argtypes = [C.c_int, C.c_double, C.c_void_p]
    L.orc_gaussian_taps_fx(n, sigma, _vp(o))
    return o


def dissolve(img1, img2, phase):
    a = np.ascontiguousarray(img1, np.uint8); b = np.ascontiguousarray(img2, np.uint8); o = np.zeros_like(a)
    L = lib(); L.orc_dissolve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    L.orc_dissolve(_vp(a), _vp(b), a.shape[1], a.shape[0], a.shape[2] if a.ndim == 3 else 1, phase, _vp(o))
    return o


def fill_convex(img, pts, value):
    m = np.ascontiguousarray(img, np.int32).copy()
    xy = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
    lib().orc_fill_convex(m.shape[1], m.shape[0], _vp(xy), len(xy), int(value), _vp(m))
    return m


def convex_hull(pts):
    p = _f(pts).reshape(-1, 2); o = np.zeros_like(p)
    n = lib().orc_convex_hull(_vp(p), len(p), _vp(o))
    return o[:n].copy()


def gabor_field(bgr):
    """gabor_filter(bgr / 255) with the default arguments (src/poppy.hpp:119-122): the direct double sum rounded once."""
    return gabor_filter_direct(u8_to_f32(np.ascontiguousarray(bgr, np.uint8)), 13, gabor_bank(13, 5, 10))


def frame_ratio(j, n, phase):
    """The scheduler of poppy::morph (src/poppy.hpp:181-210): shape (= colour) ratio of frame j."""
    linear = j / float(n)
    progress = 0.0
    if phase >= 1.0:
        progress = 1.0
    if 0 <= phase < 1.0:
        progress = 1.0 / n
    elif linear == 0:
        progress = 0.0
    elif linear == 1:
        progress = 1.0
    else:
        progress = (1.0 / (1.0 - linear)) / n
    shape = progress * phase if 0 <= phase < 1.0 else progress
    return min(shape, 1.0)


def pair_setup(img1, img2, max_keypoints=300, tolerance=1.0):
    """poppy::morph up to its frame loop (src/poppy.hpp:46-160, no face detection, no auto-align) from the raw BGR pair:
    dict(nfeatures, detail, g1, g2, kp1, kp2, points1, points2, gabor2, distance)."""
    img1 = np.ascontiguousarray(img1, np.uint8); img2 = np.ascontiguousarray(img2, np.uint8)
    h, w = img1.shape[:2]
    gf1, gf2 = foreground(img1)["foreground"], foreground(img2)["foreground"]
    d1, d2 = dft_detail2(gf1), dft_detail2(gf2)
    nfeatures = int(max_keypoints * (255.0 / max(d1, d2)))
    g1, g2 = orb_input(gf1), orb_input(gf2)
    kp1, kp2 = orb_detect(g1, nfeatures), orb_detect(g2, nfeatures)
    n = min(len(kp1), len(kp2))
    f1, f2 = filter_invalid(kp1[:n, :2].copy(), kp2[:n, :2].copy(), w, h)
    out = dict(nfeatures=nfeatures, detail=(d1, d2), g1=g1, g2=g2, kp1=kp1, kp2=kp2, gabor2=gabor_field(img2),
               points1=f1[:0], points2=f2[:0], distance=None)
    if len(f1):
        md = morph_distance(f1, f2, w, h)
        a, b = match_prepare(f1, f2, w, h, tolerance, md)
        u1 = make_uniq(clip_points(a, w, h)); u2 = make_uniq(clip_points(b, w, h))
        k = min(len(u1), len(u2))
        out.update(points1=a, points2=b, distance=morph_distance(u1[:k], u2[:k], w, h))
    return out


def morph(img1, img2, number_of_frames, phase=-1.0, levels=64, setup=None):
    """poppy::morph end to end: the list of frames it hands to the writer."""
    img1 = np.ascontiguousarray(img1, np.uint8); img2 = np.ascontiguousarray(img2, np.uint8)
    if phase == 0:
        return [img1.copy() for _ in range(number_of_frames)]
    if phase == 1:
        return [img2.copy() for _ in range(number_of_frames)]
    s = setup or pair_setup(img1, img2)
    if not len(s["points1"]):
        return [dissolve(img1, img2, phase) for _ in range(number_of_frames)]
    frames, cur, pts = [], img1, s["points1"]
    for j in range(number_of_frames):
        r = frame_ratio(j, number_of_frames, phase)
        cur, pts = morph_images(cur, img2, s["gabor2"], pts, s["points2"], r, r, levels)
        frames.append(cur)
        if phase >= 0:
            break
    return frames
