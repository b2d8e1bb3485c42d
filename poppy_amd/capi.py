"""ctypes binding of libpoppy_hip.so — mirrors include/poppy_hip.h one to one.

Importing this module never touches the GPU; Context() does and raises if there is no usable
gfx950 device or the extension is missing (no fallback path exists).
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# POPPY_HIP_LIB: another build of the same sources (tools/experiments load libpoppy_hip_experiments.so, `python -m poppy_amd.build --experiments`,
# the only build in which the result-changing measurement switches exist); tests and bench.py run on the shipped library
SO_PATH = os.environ.get("POPPY_HIP_LIB") or os.path.join(HERE, "libpoppy_hip.so")

_lib = None


class PoppySettings(C.Structure):
    _fields_ = [("number_of_frames", C.c_int), ("match_tolerance", C.c_double), ("max_keypoints", C.c_int),
                ("pyramid_levels", C.c_int), ("enable_radial_mask", C.c_int), ("enable_auto_align", C.c_int)]


WARP_KERNELS = ("k_warp_bin", "k_warp_tile", "k_warp4")        # order of poppy_hip_warp_counts
WRITE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_size_t)

# every symbol include/poppy_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "poppy_settings_default", "poppy_hip_create", "poppy_hip_destroy", "poppy_hip_last_error", "poppy_hip_create_error",
    "poppy_hip_morph_images", "poppy_hip_pair_load", "poppy_hip_pair_load_device", "poppy_hip_render", "poppy_hip_pair_reset",
    "poppy_hip_frame_device", "poppy_hip_frame_wait", "poppy_hip_frame_stream", "poppy_hip_sync", "poppy_hip_stream", "poppy_frame_ratio", "poppy_hip_morph_frames",
    "poppy_hip_dissolve", "poppy_hip_set_debug", "poppy_hip_last_warp_kind", "poppy_warp_records", "poppy_hip_hamming_knn2", "poppy_ratio_symmetry",
    "poppy_hip_pair_begin_descriptors", "poppy_hip_warp_affine", "poppy_hip_auto_align", "poppy_hip_align_step",
    "poppy_procrustes", "poppy_perspective_from4", "poppy_hip_pair_corrected2", "poppy_hip_debug_fetch", "poppy_hip_debug_triangles", "poppy_plan_frame",
    "poppy_hip_timing_summary", "poppy_hip_set_timing", "poppy_hip_render_many",
    "poppy_hip_orb_describe", "poppy_hip_hamming_match",
    "poppy_sink_open", "poppy_sink_write", "poppy_sink_close", "poppy_hip_render_phases", "poppy_hip_pool_set_timing", "poppy_hip_pool_timing_summary", "poppy_hip_pool_warp_counts", "poppy_hip_pool_create", "poppy_hip_pool_create_tuned", "poppy_hip_pool_destroy", "poppy_hip_pool_morph_pairs", "poppy_hip_pool_submit_pairs", "poppy_hip_pool_wait", "poppy_count_pair_frames_cb", "poppy_hip_warp_counts", "poppy_hip_time_last_warp", "poppy_hip_mask_rider", "poppy_hip_pool_mask_rider", "poppy_hip_comm_id", "poppy_hip_comm_init", "poppy_hip_comm_free", "poppy_hip_comm_info", "poppy_hip_pair_broadcast", "poppy_hip_comm_max",
    "poppy_hip_pair_begin_sharded", "poppy_hip_sharded_setups", "poppy_hip_pair_begin_sharded_local", "poppy_hip_pair_state_bytes", "poppy_hip_pair_export_device", "poppy_hip_pair_import_device", "poppy_hip_morph_sharded", "poppy_hip_morph_pairs",
    "poppy_dft_plan", "poppy_hip_pair_begin_device", "poppy_count_frames_cb", "poppy_hip_morph", "poppy_hip_pair_distance", "poppy_printed_morph_distance", "poppy_hypotf_selfcheck",
    "poppy_hip_orb_detect", "poppy_hip_foreground", "poppy_hip_median_blur", "poppy_match_points", "poppy_hip_pair_begin_prefiltered", "poppy_hip_pair_begin", "poppy_hip_pair_begin_info", "poppy_hip_orb_input", "poppy_hip_gabor_field", "poppy_hip_set_gabor_direct", "poppy_hip_set_setup_chains", "poppy_hip_gabor_doubt", "poppy_radial_gradient", "poppy_radial_mask", "poppy_gabor_tables", "poppy_pyr_tail_plan", "poppy_hip_blur_margin", "poppy_hip_pair_points",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is the product; there is no fallback)")
        L = C.CDLL(SO_PATH)
        L.poppy_hip_create.restype = C.c_void_p
        L.poppy_hip_create.argtypes = [C.c_int, C.POINTER(PoppySettings)]
        L.poppy_hip_destroy.argtypes = [C.c_void_p]
        L.poppy_hip_last_error.restype = C.c_char_p
        L.poppy_hip_last_error.argtypes = [C.c_void_p]
        L.poppy_hip_create_error.restype = C.c_char_p
        L.poppy_frame_ratio.restype = C.c_double
        L.poppy_frame_ratio.argtypes = [C.c_int, C.c_int, C.c_double]
        L.poppy_hip_frame_device.restype = C.c_void_p
        L.poppy_hip_frame_device.argtypes = [C.c_void_p]
        L.poppy_hip_stream.restype = C.c_void_p
        L.poppy_hip_stream.argtypes = [C.c_void_p]
        L.poppy_hip_frame_stream.restype = C.c_void_p
        L.poppy_hip_frame_stream.argtypes = [C.c_void_p]
        L.poppy_hip_frame_wait.argtypes = [C.c_void_p, C.c_void_p]
        vp, sz, i, d = C.c_void_p, C.c_size_t, C.c_int, C.c_double
        L.poppy_hip_morph_images.argtypes = [vp, vp, sz, vp, sz, vp, i, i, vp, vp, i, d, d, vp, sz, vp]
        L.poppy_hip_pair_load.argtypes = [vp, vp, sz, vp, sz, vp, i, i, vp, vp, i]
        L.poppy_hip_pair_load_device.argtypes = [vp, vp, vp, vp, i, i, vp, vp, i]
        L.poppy_hip_render.argtypes = [vp, d, d, i, vp, sz]
        L.poppy_hip_pair_reset.argtypes = [vp]
        L.poppy_hip_sync.argtypes = [vp]
        L.poppy_hip_morph_frames.argtypes = [vp, d, vp, vp]
        L.poppy_hip_dissolve.argtypes = [vp, vp, sz, vp, sz, i, i, d, vp, sz]
        L.poppy_hip_set_debug.argtypes = [vp, i]
        L.poppy_hip_last_warp_kind.argtypes = [vp]
        L.poppy_hip_set_timing.argtypes = [vp, i]
        L.poppy_hip_debug_fetch.argtypes = [vp, C.c_char_p, vp, sz]
        L.poppy_hip_debug_triangles.argtypes = [vp, vp, vp, vp, vp, i]
        L.poppy_plan_frame.argtypes = [i, i, vp, vp, i, d, i, vp, vp, vp, vp, vp, vp, vp, vp]
        L.poppy_warp_records.argtypes = [vp, vp, i, i, i, vp]
        L.poppy_hip_hamming_knn2.argtypes = [vp, vp, i, vp, i, vp]
        L.poppy_hip_warp_affine.argtypes = [vp, vp, C.c_size_t, i, i, vp, vp, C.c_size_t]
        L.poppy_hip_auto_align.argtypes = [vp, vp, C.c_size_t, i, i, vp, vp, i, vp]
        L.poppy_hip_align_step.argtypes = [vp, i, vp, C.c_size_t, i, i, vp, vp, i, vp]
        L.poppy_procrustes.argtypes = [vp, vp, i, vp, vp, vp]
        L.poppy_hip_pair_corrected2.argtypes = [vp, vp, C.c_size_t]
        L.poppy_perspective_from4.argtypes = [vp, vp, vp]
        L.poppy_ratio_symmetry.argtypes = [vp, i, vp, i, C.c_float, vp, vp]
        L.poppy_hip_pair_begin_descriptors.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, i, i, C.c_float]
        L.poppy_hip_timing_summary.argtypes = [vp, vp, vp, vp, i]
        L.poppy_hip_render_many.argtypes = [vp, vp, vp, i, i, vp, vp]
        L.poppy_hip_orb_detect.argtypes = [vp, vp, sz, i, i, i, vp, i, vp]
        L.poppy_hip_foreground.argtypes = [vp, vp, sz, i, i, vp, vp]
        L.poppy_hip_median_blur.argtypes = [vp, vp, i, i, i, i, vp]
        L.poppy_hip_comm_info.argtypes = [vp, vp, vp, vp, vp]
        L.poppy_hip_pair_begin.argtypes = [vp, vp, sz, vp, sz, i, i]
        L.poppy_hip_pair_begin_info.argtypes = [vp, vp, vp]
        L.poppy_hip_orb_input.argtypes = [vp, vp, i, i, vp, vp, vp, vp]
        L.poppy_hip_gabor_field.argtypes = [vp, vp, sz, i, i, vp]
        L.poppy_hip_set_gabor_direct.argtypes = [vp, i]
        L.poppy_hip_set_setup_chains.argtypes = [vp, i]
        L.poppy_radial_gradient.argtypes = [i, i, vp]
        L.poppy_radial_mask.argtypes = [i, i, vp]
        L.poppy_gabor_tables.argtypes = [i, vp, vp]
        L.poppy_pyr_tail_plan.argtypes = [i, i, i, i, vp, vp]
        L.poppy_hip_blur_margin.argtypes = [vp, vp, sz, i, i, i, i, vp, sz]
        L.poppy_hip_orb_describe.argtypes = [vp, vp, sz, i, i, vp, i, vp]
        L.poppy_hip_hamming_match.argtypes = [vp, vp, i, vp, i, vp, vp]
        L.poppy_match_points.argtypes = [vp, vp, i, i, i, d, vp, vp, vp, vp]
        L.poppy_hip_pair_begin_prefiltered.argtypes = [vp, vp, sz, vp, sz, vp, vp, vp, i, i, i]
        L.poppy_hip_pair_begin.argtypes = [vp, vp, sz, vp, sz, i, i]
        L.poppy_hip_pair_points.argtypes = [vp, vp, vp, i, vp]
        L.poppy_dft_plan.argtypes = [i, vp, vp, vp, vp]
        L.poppy_hip_warp_counts.argtypes = [vp, vp, vp, vp]
        L.poppy_hip_mask_rider.argtypes = [vp]
        L.poppy_hip_time_last_warp.argtypes = [vp, i, vp]
        L.poppy_hip_pool_mask_rider.argtypes = [vp]
        L.poppy_hip_pool_create.restype = C.c_void_p
        L.poppy_hip_pool_create.argtypes = [vp, i, i, vp, vp, sz]
        L.poppy_hip_pool_create_tuned.restype = C.c_void_p
        L.poppy_hip_pool_create_tuned.argtypes = [vp, i, i, vp, i, i, i, vp, vp, vp, vp, sz]
        L.poppy_hip_pool_destroy.argtypes = [vp]
        L.poppy_hip_render_phases.argtypes = [vp, vp, i, vp, vp]
        L.poppy_sink_open.restype = C.c_void_p
        L.poppy_sink_open.argtypes = [C.c_char_p, i, i, i, i, i]
        L.poppy_sink_write.argtypes = [vp, vp, i, i, sz]
        L.poppy_sink_close.argtypes = [vp]
        L.poppy_hip_pool_set_timing.argtypes = [vp, i]
        L.poppy_hip_pool_timing_summary.argtypes = [vp, vp, vp, vp, i]
        L.poppy_hip_pool_warp_counts.argtypes = [vp, vp, vp, vp]
        L.poppy_hip_pool_morph_pairs.argtypes = [vp, i, i, i, d, i, vp, vp, vp, vp, sz]
        L.poppy_hip_pool_submit_pairs.argtypes = [vp, i, i, i, d, i, vp, vp, vp]
        L.poppy_hip_pool_wait.argtypes = [vp, vp, sz]
        L.poppy_hip_comm_id.argtypes = [vp]
        L.poppy_hip_comm_init.argtypes = [vp, i, i, vp]
        L.poppy_hip_comm_free.argtypes = [vp]
        L.poppy_hip_pair_broadcast.argtypes = [vp, i, i, i]
        L.poppy_hip_comm_max.argtypes = [vp, vp]
        L.poppy_hip_pair_state_bytes.argtypes = [i, i, vp]
        L.poppy_hip_pair_export_device.argtypes = [vp, vp, sz]
        L.poppy_hip_pair_import_device.argtypes = [vp, vp, sz, i, i]
        L.poppy_hip_morph_sharded.argtypes = [vp, i, vp, vp, sz, vp, sz, i, i, i, vp, vp, vp, sz]
        L.poppy_hip_morph_pairs.argtypes = [vp, i, i, vp, i, i, i, d, vp, vp, vp, vp, sz]
        L.poppy_hip_pair_begin_device.argtypes = [vp, vp, vp, i, i]
        L.poppy_hip_pair_begin_sharded.argtypes = [vp, vp, vp, i, i, i]
        L.poppy_hip_pair_begin_sharded_local.argtypes = [C.POINTER(vp), i, vp, vp, i, i, i]
        L.poppy_hip_morph.argtypes = [vp, vp, sz, vp, sz, i, i, d, i, vp, vp, vp]
        L.poppy_hip_pair_distance.argtypes = [vp, vp]
        L.poppy_printed_morph_distance.argtypes = [vp, vp, i, i, i, vp]
        L.poppy_hypotf_selfcheck.restype = C.c_long
        L.poppy_hypotf_selfcheck.argtypes = [C.c_long, C.c_uint64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class PoppyError(RuntimeError):
    pass


def plan_frame(w, h, p1, p2, shape):
    """Host-only mesh planning (no GPU)."""
    p1 = np.ascontiguousarray(p1, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    n = len(p1)
    mt = 2 * n + 16
    nt = C.c_int(0)
    idx3 = np.zeros((mt, 3), np.int32); tri = np.zeros((mt, 3, 2), np.int32)
    M1, M2, i1, i2 = (np.zeros((mt, 3, 3), np.float32) for _ in range(4))
    mp = np.zeros((n, 2), np.float32)
    rc = lib().poppy_plan_frame(w, h, _p(p1), _p(p2), n, shape, mt, C.byref(nt), _p(idx3), _p(tri), _p(M1), _p(M2), _p(i1), _p(i2), _p(mp))
    if rc:
        raise PoppyError(f"poppy_plan_frame: {rc}")
    t = nt.value
    return dict(idx3=idx3[:t], tri_xy=tri[:t], M1=M1[:t], M2=M2[:t], inv1=i1[:t], inv2=i2[:t], morphed=mp)


def warp_records(inv1, inv2, w, h):
    """Host-only: (records (T+1, 20) float32, admitted to the tiled warp kernel?)."""
    inv1 = np.ascontiguousarray(inv1, np.float32).reshape(-1, 9)
    inv2 = np.ascontiguousarray(inv2, np.float32).reshape(-1, 9)
    t = len(inv1)
    rec = np.zeros((t + 1, 20), np.float32)
    rc = lib().poppy_warp_records(_p(inv1), _p(inv2), t, w, h, _p(rec))
    if rc < 0:
        raise PoppyError(f"poppy_warp_records: {rc}")
    return rec, bool(rc)


def ratio_symmetry(knn12, knn21, ratio=0.7):
    """Host-only ratioTest + symmetryTest (src/experiments.hpp:14-144): rows queryIdx, trainIdx, distance."""
    knn12 = np.ascontiguousarray(knn12, np.int32).reshape(-1, 4)
    knn21 = np.ascontiguousarray(knn21, np.int32).reshape(-1, 4)
    out = np.zeros((max(len(knn12), 1), 3), np.int32)
    n = C.c_int(0)
    rc = lib().poppy_ratio_symmetry(_p(knn12), len(knn12), _p(knn21), len(knn21), ratio, _p(out), C.byref(n))
    if rc:
        raise PoppyError(f"poppy_ratio_symmetry: {rc}")
    return out[:n.value]


def procrustes(x, y):
    """Host-only Procrustes(true, false)::procrustes: dict(rotation 2x2, scale, error, yprime n x 2)."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 2); y = np.ascontiguousarray(y, np.float32).reshape(-1, 2)
    rot = np.zeros((2, 2), np.float32); se = np.zeros(2, np.float32); yp = np.zeros_like(x)
    rc = lib().poppy_procrustes(_p(x), _p(y), len(x), _p(rot), _p(se), _p(yp))
    if rc:
        raise PoppyError(f"poppy_procrustes: {rc}")
    return dict(rotation=rot, scale=se[0], error=se[1], yprime=yp)


def perspective_from4(src4, dst4):
    s = np.ascontiguousarray(src4, np.float32).reshape(4, 2); d = np.ascontiguousarray(dst4, np.float32).reshape(4, 2)
    m = np.zeros((3, 3), np.float64)
    rc = lib().poppy_perspective_from4(_p(s), _p(d), _p(m))
    if rc:
        raise PoppyError(f"poppy_perspective_from4: {rc}")
    return m


def gabor_doubt():
    """(plane values near zero, near a float midpoint, pixels formed again as direct sums because of them) in the FFT form of the Gabor banks on the current device
    since the last call."""
    out = (C.c_ulonglong * 3)()
    rc = lib().poppy_hip_gabor_doubt(out)
    if rc:
        raise PoppyError(f"poppy_hip_gabor_doubt: {rc}")
    return int(out[0]), int(out[1]), int(out[2])


def gabor_tables(which):
    """(bank [16, ks, ks] float32, spectra [8, 64, 64] complex128 in the FFT kernel's position order) of the 31 / 13 tap Gabor bank (host only)."""
    bank = np.zeros((16, which, which), np.float32); spec = np.zeros((8, 64, 64, 2), np.float64)
    rc = lib().poppy_gabor_tables(int(which), _p(bank), _p(spec))
    if rc:
        raise PoppyError(f"poppy_gabor_tables: {rc}")
    return bank, spec[..., 0] + 1j * spec[..., 1]


def pyr_tail_plan(w, h, levels=64, tail_px=600):
    """(info dict, descriptors [n, 4] uint32) of the pyramid tail's tap table for a w x h frame (host only)."""
    info = (C.c_int * 6)()
    rc = lib().poppy_pyr_tail_plan(w, h, levels, tail_px, info, None)
    if rc:
        raise PoppyError(f"poppy_pyr_tail_plan: {rc}")
    desc = np.zeros((max(info[3], 1), 4), np.uint32)
    lib().poppy_pyr_tail_plan(w, h, levels, tail_px, info, _p(desc))
    keys = ("first", "wide_steps", "single_pixel_reductions", "descriptors", "lds_bytes", "ok")
    return dict(zip(keys, list(info))), desc[:info[3]]


def radial_gradient(w, h):
    """draw_radial_gradiant2 (host only)."""
    out = np.zeros((h, w), np.float32)
    rc = lib().poppy_radial_gradient(w, h, _p(out))
    if rc:
        raise PoppyError(f"poppy_radial_gradient: {rc}")
    return out


def radial_mask(w, h):
    """draw_radial_gradiant as Extractor::foreground uses it under enable_radial_mask (host only)."""
    out = np.zeros((h, w), np.float32)
    rc = lib().poppy_radial_mask(w, h, _p(out))
    if rc:
        raise PoppyError(f"poppy_radial_mask: {rc}")
    return out


WRITE_INDEXED_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_size_t)
PAIR_SOURCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
WRITE_PAIR_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_size_t)


def comm_id():
    """128 bytes to hand to every rank's Context.comm_init (ncclGetUniqueId)."""
    b = (C.c_uint8 * 128)()
    rc = lib().poppy_hip_comm_id(b)
    if rc:
        raise PoppyError(f"poppy_hip_comm_id: {rc} (librccl missing?)")
    return bytes(b)


def sharded_setups():
    """Runs of the sharded set-up protocol in this process (poppy_hip_sharded_setups)."""
    f = lib().poppy_hip_sharded_setups
    f.restype = C.c_ulonglong
    return int(f())


def pair_state_bytes(w, h):
    n = C.c_size_t(0)
    lib().poppy_hip_pair_state_bytes(w, h, C.byref(n))
    return n.value


def morph_sharded(devices, bgr1, bgr2, total_frames, collect=True, reduce=None, **settings):
    """One total_frames-frame phase-mode morph across `devices` (one process, one thread per GPU): list of frames by index.
    reduce(index, frame_view) -> what is kept instead of a copy of the frame (a digest, None ...): a 480-frame 1080p job is 3 GB of frames."""
    a = np.ascontiguousarray(bgr1, np.uint8); b = np.ascontiguousarray(bgr2, np.uint8)
    h, w = a.shape[:2]
    s = PoppySettings(); lib().poppy_settings_default(C.byref(s))
    for k, v in settings.items():
        setattr(s, k, v)
    frames = {}
    import threading
    lock = threading.Lock()

    def cb(user, idx, ptr, ww, hh, stride):
        f = np.ctypeslib.as_array(ptr, shape=(hh, stride))[:, :ww * 3].reshape(hh, ww, 3)
        f = reduce(idx, f) if reduce else f.copy()
        with lock:
            frames[idx] = f
    fn = WRITE_INDEXED_CB(cb) if collect else None
    dv = (C.c_int * len(devices))(*devices)
    err = C.create_string_buffer(512)
    rc = lib().poppy_hip_morph_sharded(dv, len(devices), C.byref(s), _p(a), w * 3, _p(b), w * 3, w, h, total_frames,
                                       C.cast(fn, C.c_void_p) if fn else None, None, err, 512)
    if rc:
        raise PoppyError(f"poppy_hip_morph_sharded: {rc}: {err.value.decode()}")
    return [frames[j] for j in sorted(frames)]


def morph_pairs(devices, pairs, contexts_per_device=2, phase=-1.0, collect=True, reduce=None, **settings):
    """`pairs` = list of (bgr1, bgr2): every pair one whole poppy::morph, spread over the devices.  Returns {pair: [frames]}
    (or the number of frames written when collect is False)."""
    pairs = [(np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)) for a, b in pairs]
    h, w = pairs[0][0].shape[:2]
    s = PoppySettings(); lib().poppy_settings_default(C.byref(s))
    for k, v in settings.items():
        setattr(s, k, v)
    out = {}
    count = [0]
    import threading
    lock = threading.Lock()

    def src(user, p, device, pa, sa, pb, sb):
        pa[0] = pairs[p][0].ctypes.data; sa[0] = w * 3
        pb[0] = pairs[p][1].ctypes.data; sb[0] = w * 3
        return 0

    def wr(user, p, j, ptr, ww, hh, stride):
        if collect:
            f = np.ctypeslib.as_array(ptr, shape=(hh, stride))[:, :ww * 3].reshape(hh, ww, 3)
            f = reduce(p, j, f) if reduce else f.copy()      # reduce(pair, index, frame_view) -> what is kept instead of a copy
            with lock:
                out.setdefault(p, {})[j] = f
        else:
            with lock:
                count[0] += 1
    fs, fw = PAIR_SOURCE_CB(src), WRITE_PAIR_CB(wr)
    dv = (C.c_int * len(devices))(*devices)
    err = C.create_string_buffer(512)
    rc = lib().poppy_hip_morph_pairs(dv, len(devices), contexts_per_device, C.byref(s), len(pairs), w, h, phase,
                                     C.cast(fs, C.c_void_p), C.cast(fw, C.c_void_p), None, err, 512)
    if rc:
        raise PoppyError(f"poppy_hip_morph_pairs: {rc}: {err.value.decode()}")
    return {p: [v[j] for j in sorted(v)] for p, v in out.items()} if collect else count[0]


def pair_begin_sharded_local(ctxs, d1, d2, w, h, root=0):
    """The sharded pair set-up between contexts of this process (context k plays rank k; d1 / d2 valid for ctxs[root]'s device)."""
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    rc = lib().poppy_hip_pair_begin_sharded_local(arr, len(ctxs), d1, d2, w, h, root)
    if rc:
        raise PoppyError(f"pair_begin_sharded_local: {rc}: " + "; ".join(lib().poppy_hip_last_error(c.h).decode() for c in ctxs))
    for c in ctxs:
        c.w, c.h_ = w, h


class Pool:
    """Persistent contexts for batches of independent pairs (poppy_hip_pool_*)."""

    def __init__(self, devices, contexts_per_device=2, tuned_for=None, max_candidates=3, **settings):
        """tuned_for=(width, height): poppy_hip_pool_create_tuned — the library makes up to max_candidates pools, times a calibration batch on each and keeps
        the fastest (self.candidates_ms: every pool's batch time in the order made, self.kept: the index kept)."""
        s = PoppySettings(); lib().poppy_settings_default(C.byref(s))
        for k, v in settings.items():
            setattr(s, k, v)
        dv = (C.c_int * len(devices))(*devices)
        err = C.create_string_buffer(512)
        self.candidates_ms, self.kept = None, None
        if tuned_for is None:
            self.h = lib().poppy_hip_pool_create(dv, len(devices), contexts_per_device, C.byref(s), err, 512)
        else:
            ms = (C.c_float * max_candidates)(); n = C.c_int(0); kept = C.c_int(-1)
            self.h = lib().poppy_hip_pool_create_tuned(dv, len(devices), contexts_per_device, C.byref(s), int(tuned_for[0]), int(tuned_for[1]), max_candidates,
                                                       ms, C.byref(n), C.byref(kept), err, 512)
            self.candidates_ms, self.kept = [float(ms[k]) for k in range(n.value)], kept.value
        if not self.h:
            raise PoppyError("poppy_hip_pool_create: " + err.value.decode())

    def close(self):
        if self.h:
            lib().poppy_hip_pool_destroy(self.h)
            self.h = None

    def set_timing(self, on):
        lib().poppy_hip_pool_set_timing(self.h, int(on))

    def timing_summary(self):
        names = (C.c_char_p * 32)(); ms = (C.c_float * 32)(); cnt = (C.c_int * 32)()
        n = lib().poppy_hip_pool_timing_summary(self.h, names, ms, cnt, 32)
        return [(names[k].decode(), ms[k], cnt[k]) for k in range(n)]

    def warp_counts(self):
        f, a, b = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
        lib().poppy_hip_pool_warp_counts(self.h, C.byref(f), C.byref(a), C.byref(b))
        return f.value, a.value, b.value

    def mask_rider(self):
        """True when the warp kernel also writes lbmask (else the level-0 blend kernels compute it from m2)."""
        return lib().poppy_hip_pool_mask_rider(self.h) == 1

    def warp_kernel_name(self):
        n = self.warp_counts()
        return WARP_KERNELS[n.index(max(n))]

    def morph_pairs_device_counted(self, ptr_pairs, w, h, phase=-1.0):
        """ptr_pairs: list of (device pointer of image 1, of image 2) on the pool's (single) device; frames go to the counting
        writer.  Returns the number of frames written."""
        def src(user, p, device, pa, sa, pb, sb):
            pa[0] = ptr_pairs[p][0]; sa[0] = w * 3
            pb[0] = ptr_pairs[p][1]; sb[0] = w * 3
            return 0
        fs = PAIR_SOURCE_CB(src)
        n = C.c_longlong(0)
        err = C.create_string_buffer(512)
        rc = lib().poppy_hip_pool_morph_pairs(self.h, len(ptr_pairs), w, h, phase, 1, C.cast(fs, C.c_void_p),
                                              C.cast(lib().poppy_count_pair_frames_cb, C.c_void_p), C.cast(C.byref(n), C.c_void_p), err, 512)
        if rc:
            raise PoppyError(f"poppy_hip_pool_morph_pairs: {rc}: {err.value.decode()}")
        return n.value

    # ---- batches without waiting for them (poppy_hip_pool_submit_pairs / poppy_hip_pool_wait) ----
    def submit_pairs_device_counted(self, ptr_pairs, w, h, phase=-1.0, on_device=True):
        """Queues the batch and returns; wait() returns the frames written by every batch submitted since the last wait().  on_device=False: the pointers are
        HOST pointers (tight rows; pinned memory if the uploads are to overlap anything) and every pair goes through poppy_hip_morph."""
        ptr_pairs = list(ptr_pairs)
        def src(user, p, device, pa, sa, pb, sb):
            pa[0] = ptr_pairs[p][0]; sa[0] = w * 3
            pb[0] = ptr_pairs[p][1]; sb[0] = w * 3
            return 0
        fs = PAIR_SOURCE_CB(src)
        if not hasattr(self, "_pending"):
            self._pending, self._count = [], C.c_longlong(0)
        self._pending.append((fs, ptr_pairs))                     # the callbacks stay alive until the wait
        rc = lib().poppy_hip_pool_submit_pairs(self.h, len(ptr_pairs), w, h, phase, 1 if on_device else 0, C.cast(fs, C.c_void_p),
                                               C.cast(lib().poppy_count_pair_frames_cb, C.c_void_p), C.cast(C.byref(self._count), C.c_void_p))
        if rc:
            raise PoppyError(f"poppy_hip_pool_submit_pairs: {rc}")

    def submit_pairs(self, pairs, write, phase=-1.0, inputs_on_device=False, w=None, h=None):
        """pairs: list of (bgr1, bgr2) host arrays — or (pointer 1, pointer 2) device pointers with inputs_on_device, w, h; write(pair, frame, view) is called
        from the pool's threads.  Returns at once; wait() returns when everything submitted has been written."""
        if not inputs_on_device:
            pairs = [(np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)) for a, b in pairs]
            h, w = pairs[0][0].shape[:2]
        def src(user, p, device, pa, sa, pb, sb):
            pa[0] = pairs[p][0] if inputs_on_device else pairs[p][0].ctypes.data; sa[0] = w * 3
            pb[0] = pairs[p][1] if inputs_on_device else pairs[p][1].ctypes.data; sb[0] = w * 3
            return 0
        def wr(user, p, j, ptr, ww, hh, stride):
            write(p, j, np.ctypeslib.as_array(ptr, shape=(hh, stride))[:, :ww * 3].reshape(hh, ww, 3))
        fs, fw = PAIR_SOURCE_CB(src), WRITE_PAIR_CB(wr)
        if not hasattr(self, "_pending"):
            self._pending, self._count = [], C.c_longlong(0)
        self._pending.append((fs, fw, pairs))
        rc = lib().poppy_hip_pool_submit_pairs(self.h, len(pairs), w, h, phase, 1 if inputs_on_device else 0, C.cast(fs, C.c_void_p), C.cast(fw, C.c_void_p), None)
        if rc:
            raise PoppyError(f"poppy_hip_pool_submit_pairs: {rc}")

    def wait(self):
        """Every batch submitted so far has been rendered and written; returns the frames the counting writer saw since the last wait()."""
        err = C.create_string_buffer(512)
        rc = lib().poppy_hip_pool_wait(self.h, err, 512)
        n = 0
        if hasattr(self, "_pending"):
            n, self._count.value = self._count.value, 0
            self._pending.clear()
        if rc:
            raise PoppyError(f"poppy_hip_pool_wait: {rc}: {err.value.decode()}")
        return n


def dft_plan(n):
    """Host-only: (factors, itab, wave) of the length-n transform plan."""
    f = np.zeros(34, np.int32); nf = C.c_int(0); itab = np.zeros(n, np.int32); wave = np.zeros((n, 2), np.float32)
    rc = lib().poppy_dft_plan(n, _p(f), C.byref(nf), _p(itab), _p(wave))
    if rc:
        raise PoppyError(f"poppy_dft_plan: {rc}")
    return f[:nf.value].copy(), itab, wave


def printed_morph_distance(p1, p2, w, h):
    """Host-only: the "morph distance" poppy::morph prints for prepared point lists (src/poppy.hpp:142-159)."""
    p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
    d = C.c_double(0)
    rc = lib().poppy_printed_morph_distance(_p(p1), _p(p2), len(p1), w, h, C.byref(d))
    if rc:
        raise PoppyError(f"poppy_printed_morph_distance: {rc}")
    return d.value


def match_points(p1, p2, w, h, tolerance=1.0):
    """Host-only matcher (no GPU): returns (points1, points2, initial_morph_distance)."""
    p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
    n = len(p1)
    o1 = np.zeros((n + 4, 2), np.float32); o2 = np.zeros((n + 4, 2), np.float32)
    m = C.c_int(0); imd = C.c_double(0)
    rc = lib().poppy_match_points(_p(p1), _p(p2), n, w, h, tolerance, _p(o1), _p(o2), C.byref(m), C.byref(imd))
    if rc:
        raise PoppyError(f"poppy_match_points: {rc}")
    return o1[:m.value].copy(), o2[:m.value].copy(), imd.value


class Context:
    def __init__(self, device=0, **settings):
        L = lib()
        s = PoppySettings()
        L.poppy_settings_default(C.byref(s))
        for k, v in settings.items():
            setattr(s, k, v)
        self.settings = s
        self.h = L.poppy_hip_create(device, C.byref(s))
        if not self.h:
            raise PoppyError("poppy_hip_create failed: " + L.poppy_hip_create_error().decode())
        self.w = self.h_ = 0

    def close(self):
        if self.h:
            lib().poppy_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise PoppyError(f"{what}: {rc}: {lib().poppy_hip_last_error(self.h).decode()}")

    def morph_images(self, c1, c2, gabor2, p1, p2, shape, mask):
        c1 = np.ascontiguousarray(c1, np.uint8); c2 = np.ascontiguousarray(c2, np.uint8)
        g = np.ascontiguousarray(gabor2, np.float32)
        p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
        h, w = c1.shape[:2]
        out = np.empty((h, w, 3), np.uint8)
        mp = np.empty((len(p1), 2), np.float32)
        self._chk(lib().poppy_hip_morph_images(self.h, _p(c1), w * 3, _p(c2), w * 3, _p(g), w, h, _p(p1), _p(p2), len(p1),
                                               shape, mask, _p(out), w * 3, _p(mp)), "morph_images")
        self.w, self.h_ = w, h
        return out, mp

    def pair_load(self, c1, c2, gabor2, p1, p2):
        c1 = np.ascontiguousarray(c1, np.uint8); c2 = np.ascontiguousarray(c2, np.uint8)
        g = np.ascontiguousarray(gabor2, np.float32)
        p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
        h, w = c1.shape[:2]
        self._chk(lib().poppy_hip_pair_load(self.h, _p(c1), w * 3, _p(c2), w * 3, _p(g), w, h, _p(p1), _p(p2), len(p1)), "pair_load")
        self.w, self.h_ = w, h

    def pair_load_device(self, d1, d2, dg, w, h, p1, p2):
        p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
        self._chk(lib().poppy_hip_pair_load_device(self.h, d1, d2, dg, w, h, _p(p1), _p(p2), len(p1)), "pair_load_device")
        self.w, self.h_ = w, h

    def render(self, shape, mask, chain=False, fetch=True):
        out = np.empty((self.h_, self.w, 3), np.uint8) if fetch else None
        self._chk(lib().poppy_hip_render(self.h, shape, mask, int(chain), _p(out), self.w * 3), "render")
        return out

    def orb_detect(self, gray, nfeatures):
        g = np.ascontiguousarray(gray, np.uint8)
        h, w = g.shape
        cap = max(2 * nfeatures + 64, 64)
        kp = np.zeros((cap, 7), np.float32)
        n = C.c_int(0)
        self._chk(lib().poppy_hip_orb_detect(self.h, _p(g), w, w, h, nfeatures, _p(kp), cap, C.byref(n)), "orb_detect")
        return kp[:n.value].copy()

    def median_blur(self, img, ksize, form=0):
        """cv::medianBlur on a single-channel u8 image; form: 0 chain default, 1 lane per column, 2 column histograms over the ranks of each tile's values,
        3 the same without presence maps, 4 no one-count tiles, 5 every tile by windows of ranks, 6 / 7 first window at the top / bottom (include/poppy_hip.h)."""
        a = np.ascontiguousarray(img, np.uint8)
        h, w = a.shape
        out = np.zeros((h, w), np.uint8)
        self._chk(lib().poppy_hip_median_blur(self.h, _p(a), w, h, int(ksize), int(form), _p(out)), "median_blur")
        return out

    def foreground(self, bgr, debug=False):
        """Extractor::foreground for one BGR image -> goodFeatures (u8); with debug=True a dict with every intermediate."""
        a = np.ascontiguousarray(bgr, np.uint8)
        h, w = a.shape[:2]
        out = np.zeros((h, w), np.uint8)
        if not debug:
            self._chk(lib().poppy_hip_foreground(self.h, _p(a), w * 3, w, h, _p(out), None), "foreground")
            return out
        grey = np.zeros((h, w), np.uint8); masked = np.zeros((h, w), np.uint8)
        stages = np.zeros((50, h, w), np.uint8); floats = np.zeros((3, h, w), np.float32)

        class Dbg(C.Structure):
            _fields_ = [("grey", C.c_void_p), ("stages", C.c_void_p), ("floats", C.c_void_p), ("masked", C.c_void_p)]
        d = Dbg(grey.ctypes.data, stages.ctypes.data, floats.ctypes.data, masked.ctypes.data)
        self._chk(lib().poppy_hip_foreground(self.h, _p(a), w * 3, w, h, _p(out), C.byref(d)), "foreground")
        res = dict(foreground=out, grey=grey, masked=masked, lin=floats[0], logged=floats[1], finalMask=floats[2],
                   flow0=stages[0], acc0=stages[1])
        for k in range(12):
            res[f"med{k + 1}"], res[f"flow{k + 1}"], res[f"acc{k + 1}"], res[f"blur{k + 1}"] = stages[2 + 4 * k: 6 + 4 * k]
        return res

    def pair_begin(self, bgr1, bgr2):
        """Raw BGR pair -> resident pair (pre-ORB chain, ORB, matcher, gabor2 on the GPU). Returns (nfeatures, (d1, d2))."""
        a = np.ascontiguousarray(bgr1, np.uint8); b = np.ascontiguousarray(bgr2, np.uint8)
        h, w = a.shape[:2]
        self._chk(lib().poppy_hip_pair_begin(self.h, _p(a), w * 3, _p(b), w * 3, w, h), "pair_begin")
        self.w, self.h_ = w, h
        nf = C.c_int(0); d = (C.c_double * 2)()
        lib().poppy_hip_pair_begin_info(self.h, C.byref(nf), d)
        return nf.value, (d[0], d[1])

    def pair_begin_info(self):
        """(nfeatures, (detail of image 1, of image 2)) of the resident pair's set-up."""
        nf = C.c_int(0); d = (C.c_double * 2)()
        lib().poppy_hip_pair_begin_info(self.h, C.byref(nf), d)
        return nf.value, (d[0], d[1])

    def comm_init(self, rank, world, id128):
        buf = (C.c_uint8 * 128).from_buffer_copy(id128)
        self._chk(lib().poppy_hip_comm_init(self.h, rank, world, buf), "comm_init")

    def comm_info(self):
        """(rank, world) as given to comm_init and (rank, count) as the RCCL communicator reports them (-1 without one)."""
        v = [C.c_int(-1) for _ in range(4)]
        self._chk(lib().poppy_hip_comm_info(self.h, *[C.byref(x) for x in v]), "comm_info")
        return tuple(x.value for x in v)

    def comm_free(self):
        lib().poppy_hip_comm_free(self.h)

    def pair_broadcast(self, root, w, h):
        self._chk(lib().poppy_hip_pair_broadcast(self.h, root, w, h), "pair_broadcast")
        self.w, self.h_ = w, h

    def comm_max(self, value):
        d = C.c_double(value)
        self._chk(lib().poppy_hip_comm_max(self.h, C.byref(d)), "comm_max")
        return d.value

    def pair_export_device(self, d_dst, nbytes):
        self._chk(lib().poppy_hip_pair_export_device(self.h, d_dst, nbytes), "pair_export_device")

    def pair_import_device(self, d_src, nbytes, w, h):
        self._chk(lib().poppy_hip_pair_import_device(self.h, d_src, nbytes, w, h), "pair_import_device")
        self.w, self.h_ = w, h

    def pair_begin_sharded(self, d1, d2, w, h, root=0):
        """Collective over this context's communicator: the pair set-up itself spread over the ranks (d1 / d2: device pointers of the raw
        pair on rank `root`, None elsewhere); every rank ends up with the resident pair."""
        self._chk(lib().poppy_hip_pair_begin_sharded(self.h, d1, d2, w, h, root), "pair_begin_sharded")
        self.w, self.h_ = w, h

    def pair_begin_device(self, d1, d2, w, h):
        """Raw BGR pair already in this GPU's memory (device pointers, tight rows)."""
        self._chk(lib().poppy_hip_pair_begin_device(self.h, d1, d2, w, h), "pair_begin_device")
        self.w, self.h_ = w, h

    def morph_frames_counted(self, phase=-1.0):
        """The frame loop with every frame handed to the library's counting writer (pinned host hand-off, no Python per frame)."""
        n = C.c_longlong(0)
        cb = C.cast(lib().poppy_count_frames_cb, C.c_void_p)
        self._chk(lib().poppy_hip_morph_frames(self.h, phase, cb, C.cast(C.byref(n), C.c_void_p)), "morph_frames")
        return n.value

    def render_phases(self, ts, write=None, counted=False):
        """Frames of the sharded job: phase t_k each (t == 0 / 1: copies of image 1 / 2).  counted: frames go to the library's
        counting writer and the count is returned."""
        t = np.ascontiguousarray(ts, np.float64)
        if counted:
            n = C.c_longlong(0)
            self._chk(lib().poppy_hip_render_phases(self.h, _p(t), len(t), C.cast(lib().poppy_count_frames_cb, C.c_void_p), C.cast(C.byref(n), C.c_void_p)), "render_phases")
            return n.value
        fn = None
        if write is not None:
            def cb(user, ptr, w, h, stride):
                write(np.ctypeslib.as_array(ptr, shape=(h, stride))[:, :w * 3].reshape(h, w, 3))
            fn = WRITE_CB(cb)
        self._chk(lib().poppy_hip_render_phases(self.h, _p(t), len(t), C.cast(fn, C.c_void_p) if fn else None, None), "render_phases")

    def render_many_counted(self, shapes, chain=False):
        sh = np.ascontiguousarray(shapes, np.float64)
        n = C.c_longlong(0)
        cb = C.cast(lib().poppy_count_frames_cb, C.c_void_p)
        self._chk(lib().poppy_hip_render_many(self.h, _p(sh), _p(sh), len(sh), int(chain), cb, C.cast(C.byref(n), C.c_void_p)), "render_many")
        return n.value

    def orb_input(self, good_features):
        gf = np.ascontiguousarray(good_features, np.uint8)
        h, w = gf.shape
        g = np.zeros((h, w), np.uint8); us = np.zeros((h, w), np.float32); gb = np.zeros((h, w), np.float32); d = C.c_double(0)
        self._chk(lib().poppy_hip_orb_input(self.h, _p(gf), w, h, _p(g), _p(us), _p(gb), C.byref(d)), "orb_input")
        return dict(g=g, us=us, gb=gb, detail=d.value)

    def blur_margin(self, bgr, union_w, union_h):
        a = np.ascontiguousarray(bgr, np.uint8)
        h, w = a.shape[:2]
        out = np.zeros((union_h, union_w, 3), np.uint8)
        self._chk(lib().poppy_hip_blur_margin(self.h, _p(a), w * 3, w, h, union_w, union_h, _p(out), union_w * 3), "blur_margin")
        return out

    def time_last_warp(self, reps=50):
        """Average milliseconds per launch of the last frame's fused raster + warp kernel, relaunched back to back, nothing else running."""
        ms = C.c_float(0)
        self._chk(lib().poppy_hip_time_last_warp(self.h, int(reps), C.byref(ms)), "time_last_warp")
        return ms.value

    def set_gabor_direct(self, on=True):
        """Gabor banks as direct double sums (True) or tiled FFTs (False, the default)."""
        self._chk(lib().poppy_hip_set_gabor_direct(self.h, int(on)), "set_gabor_direct")

    def set_setup_chains(self, serial=True):
        """pair set-up: the two images' chains one after the other (True; what a pool of >= 3 contexts uses) or side by side (False, a context's default)."""
        self._chk(lib().poppy_hip_set_setup_chains(self.h, int(serial)), "set_setup_chains")

    def gabor_field(self, bgr):
        a = np.ascontiguousarray(bgr, np.uint8)
        h, w = a.shape[:2]
        out = np.zeros((h, w, 3), np.float32)
        self._chk(lib().poppy_hip_gabor_field(self.h, _p(a), w * 3, w, h, _p(out)), "gabor_field")
        return out

    def orb_describe(self, gray, kps7):
        g = np.ascontiguousarray(gray, np.uint8)
        k = np.ascontiguousarray(kps7, np.float32)
        h, w = g.shape
        out = np.zeros((len(k), 32), np.uint8)
        self._chk(lib().poppy_hip_orb_describe(self.h, _p(g), w, w, h, _p(k), len(k), _p(out)), "orb_describe")
        return out

    def hamming_match(self, query, train):
        q = np.ascontiguousarray(query, np.uint8); t = np.ascontiguousarray(train, np.uint8)
        out = np.zeros((len(q), 3), np.int32)
        n = C.c_int(0)
        self._chk(lib().poppy_hip_hamming_match(self.h, _p(q), len(q), _p(t), len(t), _p(out), C.byref(n)), "hamming_match")
        return out[:n.value].copy()

    def pair_corrected2(self, w, h):
        out = np.zeros((h, w, 3), np.uint8)
        self._chk(lib().poppy_hip_pair_corrected2(self.h, _p(out), w * 3), "pair_corrected2")
        return out

    def warp_affine(self, img, M):
        a = np.ascontiguousarray(img, np.uint8)
        h, w = a.shape[:2]
        m = np.ascontiguousarray(M, np.float64).reshape(6)
        out = np.zeros_like(a)
        self._chk(lib().poppy_hip_warp_affine(self.h, _p(a), w * 3, w, h, _p(m), _p(out), w * 3), "warp_affine")
        return out

    def align(self, which, img, p1, p2):
        """which: 'retranslate' | 'reprocrustes' | 'rerotate' | 'auto' -> (image, points2, distance)."""
        im = np.ascontiguousarray(img, np.uint8).copy()
        a = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); b = np.ascontiguousarray(p2, np.float32).reshape(-1, 2).copy()
        h, w = im.shape[:2]
        d = C.c_double(0)
        if which == "auto":
            self._chk(lib().poppy_hip_auto_align(self.h, _p(im), w * 3, w, h, _p(a), _p(b), len(a), C.byref(d)), "auto_align")
        else:
            code = {"retranslate": 0, "reprocrustes": 1, "rerotate": 2}[which]
            self._chk(lib().poppy_hip_align_step(self.h, code, _p(im), w * 3, w, h, _p(a), _p(b), len(a), C.byref(d)), "align_step")
        return im, b, d.value

    def hamming_knn2(self, query, train):
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        out = np.zeros((len(q), 4), np.int32)
        self._chk(lib().poppy_hip_hamming_knn2(self.h, _p(q), len(q), _p(t), len(t), _p(out)), "hamming_knn2")
        return out

    def pair_begin_descriptors(self, bgr1, bgr2, ratio=0.7):
        """Opt-in descriptor mode of the pair set-up; returns nfeatures."""
        a = np.ascontiguousarray(bgr1, np.uint8); b = np.ascontiguousarray(bgr2, np.uint8)
        h, w = a.shape[:2]
        self._chk(lib().poppy_hip_pair_begin_descriptors(self.h, _p(a), w * 3, _p(b), w * 3, w, h, ratio), "pair_begin_descriptors")
        nf = C.c_int(0)
        lib().poppy_hip_pair_begin_info(self.h, C.byref(nf), None)
        return nf.value

    def pair_begin_prefiltered(self, bgr1, bgr2, g1, g2, gabor2, nfeatures):
        a = np.ascontiguousarray(bgr1, np.uint8); b = np.ascontiguousarray(bgr2, np.uint8)
        g1 = np.ascontiguousarray(g1, np.uint8); g2 = np.ascontiguousarray(g2, np.uint8)
        g = np.ascontiguousarray(gabor2, np.float32)
        h, w = a.shape[:2]
        self._chk(lib().poppy_hip_pair_begin_prefiltered(self.h, _p(a), w * 3, _p(b), w * 3, _p(g1), _p(g2), _p(g), w, h, nfeatures), "pair_begin_prefiltered")
        self.w, self.h_ = w, h

    def pair_points(self, max_points=8192):
        p1 = np.zeros((max_points, 2), np.float32); p2 = np.zeros((max_points, 2), np.float32)
        n = C.c_int(0)
        self._chk(lib().poppy_hip_pair_points(self.h, _p(p1), _p(p2), max_points, C.byref(n)), "pair_points")
        return p1[:n.value].copy(), p2[:n.value].copy()

    def morph(self, bgr1, bgr2, phase=-1.0, distance=False, collect=True, row_pad=(0, 0)):
        """poppy::morph end to end: returns (status, frames, printed morph distance or None).  status is POPPY_OK (0) or
        POPPY_E_NOMATCH (-5, fallback frames written); anything else raises.  row_pad = (bytes, bytes): the two images are handed over with that many
        extra bytes per row (cv::Mat ROIs, src/poppy.cpp:234-239: step > cols * 3), the padding filled with a pattern no pixel may come from."""
        a = np.ascontiguousarray(bgr1, np.uint8); b = np.ascontiguousarray(bgr2, np.uint8)
        h, w = a.shape[:2]
        strides = [w * 3 + int(row_pad[0]), w * 3 + int(row_pad[1])]
        if row_pad != (0, 0):
            padded = []
            for img, st in zip((a, b), strides):
                buf = np.full((h, st), 0xA5, np.uint8)
                buf[:, :w * 3] = img.reshape(h, w * 3)
                padded.append(buf)
            a, b = padded
        frames = []

        def cb(user, ptr, ww, hh, stride):
            frames.append(np.ctypeslib.as_array(ptr, shape=(hh, stride))[:, :ww * 3].reshape(hh, ww, 3).copy())
        fn = WRITE_CB(cb) if collect else None
        d = C.c_double(float("nan"))
        rc = lib().poppy_hip_morph(self.h, _p(a), strides[0], _p(b), strides[1], w, h, phase, int(distance),
                                   C.cast(fn, C.c_void_p) if fn else None, None, C.byref(d))
        if rc not in (0, -5):
            self._chk(rc, "morph")
        self.w, self.h_ = w, h
        return rc, frames, (None if d.value != d.value else d.value)

    def pair_distance(self):
        d = C.c_double(0)
        self._chk(lib().poppy_hip_pair_distance(self.h, C.byref(d)), "pair_distance")
        return d.value

    def reset(self):
        self._chk(lib().poppy_hip_pair_reset(self.h), "pair_reset")

    def sync(self):
        self._chk(lib().poppy_hip_sync(self.h), "sync")

    def morph_frames(self, phase=-1.0, collect=True):
        frames = []

        def cb(user, ptr, w, h, stride):
            frames.append(np.ctypeslib.as_array(ptr, shape=(h, stride))[:, :w * 3].reshape(h, w, 3).copy())
        fn = WRITE_CB(cb) if collect else None
        self._chk(lib().poppy_hip_morph_frames(self.h, phase, C.cast(fn, C.c_void_p) if fn else None, None), "morph_frames")
        return frames

    def dissolve(self, img1, img2, phase):
        a = np.ascontiguousarray(img1, np.uint8); b = np.ascontiguousarray(img2, np.uint8)
        h, w = a.shape[:2]
        out = np.empty((h, w, 3), np.uint8)
        self._chk(lib().poppy_hip_dissolve(self.h, _p(a), w * 3, _p(b), w * 3, w, h, phase, _p(out), w * 3), "dissolve")
        return out

    def warp_counts(self):
        f, a, b = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
        lib().poppy_hip_warp_counts(self.h, C.byref(f), C.byref(a), C.byref(b))
        return f.value, a.value, b.value

    def mask_rider(self):
        """True when the warp kernel also writes lbmask (else the level-0 blend kernels compute it from m2)."""
        return lib().poppy_hip_mask_rider(self.h) == 1

    def warp_kernel_name(self):
        n = self.warp_counts()
        return WARP_KERNELS[n.index(max(n))]

    def last_warp_kind(self):
        return int(lib().poppy_hip_last_warp_kind(self.h))

    def set_debug(self, on=True):
        lib().poppy_hip_set_debug(self.h, int(on))

    def set_timing(self, on=1):
        """0 off, 1 every kernel group (frames issued launch by launch), 2 the map+remap kernel only."""
        lib().poppy_hip_set_timing(self.h, int(on))

    def fetch(self, name):
        h, w = self.h_, self.w
        shapes = {"gabor2": ((h, w, 3), np.float32), "triMap": ((h, w), np.int32), "trImg1": ((h, w, 3), np.uint8), "trImg2": ((h, w, 3), np.uint8),
                  "lbmask": ((h, w), np.float32), "m2": ((h, w), np.float32), "lapBlend": ((h, w, 3), np.float32),
                  "unsharp": ((h, w, 3), np.float32)}
        shp, dt = shapes[name]
        out = np.empty(shp, dt)
        self._chk(lib().poppy_hip_debug_fetch(self.h, name.encode(), _p(out), out.nbytes), "debug_fetch " + name)
        return out

    def triangles(self, max_tris=8192):
        nt = C.c_int(0)
        idx3 = np.zeros((max_tris, 3), np.int32)
        M1 = np.zeros((max_tris, 3, 3), np.float32); M2 = np.zeros((max_tris, 3, 3), np.float32)
        self._chk(lib().poppy_hip_debug_triangles(self.h, C.byref(nt), _p(idx3), _p(M1), _p(M2), max_tris), "debug_triangles")
        t = nt.value
        return idx3[:t], M1[:t], M2[:t]

    def timing_summary(self):
        """[(kernel group, total ms, launches)] since timing was switched on / last summary (drains the stream)."""
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        cnt = (C.c_int * 32)()
        n = lib().poppy_hip_timing_summary(self.h, names, ms, cnt, 32)
        return [(names[i].decode(), ms[i], cnt[i]) for i in range(n)]

    def render_many(self, shapes, masks=None, chain=False, write=None):
        """write: optional callable(frame_view) invoked for every frame with a view of the pinned host copy (valid during the call)."""
        sh = np.ascontiguousarray(shapes, np.float64)
        mk = sh if masks is None else np.ascontiguousarray(masks, np.float64)
        fn = None
        if write is not None:
            def cb(user, ptr, w, h, stride):
                write(np.ctypeslib.as_array(ptr, shape=(h, stride))[:, :w * 3].reshape(h, w, 3))
            fn = WRITE_CB(cb)
        self._chk(lib().poppy_hip_render_many(self.h, _p(sh), _p(mk), len(sh), int(chain), C.cast(fn, C.c_void_p) if fn else None, None), "render_many")

    def frame_device_ptr(self):
        return lib().poppy_hip_frame_device(self.h)

    def stream_ptr(self):
        return lib().poppy_hip_stream(self.h)

    def frame_stream_ptr(self):
        return lib().poppy_hip_frame_stream(self.h)

    def frame_wait(self, stream_ptr=None):
        """Orders `stream_ptr` (a hipStream_t; None: the host) behind the last frame rendered (phase-mode frames run on streams of their own)."""
        self._chk(lib().poppy_hip_frame_wait(self.h, stream_ptr), "frame_wait")
