"""GPU parity on whole calls of poppy::morph through the C ABI (poppy_hip_morph), against frames of the REAL reference
(tests/golden, captured from the compiled poppy::morph): every BASELINE.json single-GPU configuration end to end from the raw
image pair, the phase == 0 / 1 short-circuits, --distance, the no-match fallback expression and shallow pyramids."""
import os

import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


def _ctx(**kw):
    from poppy_amd import capi
    return capi.Context(0, **kw)


def test_phase_mode_frame_fixture():
    """a_256x256_phase: init(numberOfFrames = 1); morph(phase = 0.5) of the real reference."""
    case = "a_256x256_phase"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=int(inp["cfg"][0]))
    rc, frames, dist = c.morph(inp["img1"], inp["img2"], phase=float(inp["cfg"][1]))
    assert rc == 0 and len(frames) == 1
    G.check(case, "frame0", frames[0])
    c.close()


def test_phase_zero_and_one_short_circuit():
    case = "a_256x256_phase01"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=3)
    rc, frames, _ = c.morph(inp["img1"], inp["img2"], phase=0.0)
    assert rc == 0 and len(frames) == 3
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
    c.close()
    c = _ctx(number_of_frames=1)
    rc, frames, _ = c.morph(inp["img1"], inp["img2"], phase=1.0)
    assert rc == 0 and len(frames) == 1
    G.check(case, "phase0_frame", frames[0])
    # the same two shortcuts on a resident pair (what a rank of the sharded job calls for t_0 = 0)
    c.pair_begin(inp["img1"], inp["img2"])
    f0 = c.morph_frames(0.0)
    f1 = c.morph_frames(1.0)
    assert len(f0) == 1 and np.array_equal(f0[0], inp["img1"]) and np.array_equal(f1[0], inp["img2"])
    c.close()


def test_cfg1_512x512_30_chained_frames():
    """BASELINE.json configs[0] end to end from the raw pair: all 30 frames of the real poppy::morph (sha256)."""
    case = "a_512x512_chain30"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=30)
    rc, frames, dist = c.morph(inp["img1"], inp["img2"])
    assert rc == 0 and len(frames) == 30
    assert dist == float(G.full(case, "printedMorphDist")[0])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1); G.check(case, "prepared2", p2)
    for j, f in enumerate(frames):
        G.check(case, f"frame{j}", f)
    c.close()


def test_cfg2_1080p_60_chained_frames_and_phase_frames():
    """BASELINE.json configs[1] end to end from the raw 1080p pair: nfeatures, point pairs, the printed morph distance, all 60
    chained frames, and two phase-mode frames (t = 0.25, 0.5 with number_of_frames = 1: what a rank of configs[3] renders)."""
    case = "a_1920x1080_chain60"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=60)
    rc, frames, dist = c.morph(inp["img1"], inp["img2"])
    assert rc == 0 and len(frames) == 60
    ref = G.full(case, "detail")
    nf = __import__("ctypes").c_int(0)
    from poppy_amd import capi
    capi.lib().poppy_hip_pair_begin_info(c.h, __import__("ctypes").byref(nf), None)
    assert nf.value == int(ref[3])
    assert dist == float(G.full(case, "printedMorphDist")[0])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1); G.check(case, "prepared2", p2)
    bad = [j for j, f in enumerate(frames) if G.sha(f) != G.entries(case)[f"frame{j}"]["sha256"]]
    assert not bad, f"frames {bad} differ from the reference"
    c.close()
    c1 = _ctx(number_of_frames=1)
    for k, t in enumerate(inp["cfg"][4:]):
        rc, fr, _ = c1.morph(inp["img1"], inp["img2"], phase=float(t))
        assert rc == 0 and len(fr) == 1
        G.check(case, f"phase{k}_frame", fr[0], what=f"phase {t}")
    # the sharded job renders the same frames from ONE resident pair with explicit ratios (poppy_hip_render_many)
    got = []
    c1.reset()
    c1.render_many(np.array(inp["cfg"][4:], np.float64), chain=False, write=lambda f: got.append(f.copy()))
    for k in range(len(got)):
        G.check(case, f"phase{k}_frame", got[k])
    c1.close()


def test_cfg2_1080p_60_chained_frames_of_the_sample_photographs():
    """BASELINE.json configs[1] on content that is not synthetic: the reference's own sample photographs (images/amir1.jpg / amir2.jpg, committed as pixels, upscaled to
    1920 x 1080 in integers) through the whole of poppy::morph — nfeatures, the prepared point pairs, the printed morph distance and all 60 chained frames against a run
    of the real reference on the same pixels (fixture a_1920x1080_photo60, frames by sha256)."""
    case = "a_1920x1080_photo60"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=60)
    rc, frames, dist = c.morph(inp["img1"], inp["img2"])
    assert rc == 0 and len(frames) == 60
    nf, det = c.pair_begin_info()
    ref = G.full(case, "detail")
    assert nf == int(ref[3]) and det == (ref[0], ref[1])
    assert dist == float(G.full(case, "printedMorphDist")[0])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1); G.check(case, "prepared2", p2)
    bad = [j for j, f in enumerate(frames) if G.sha(f) != G.entries(case)[f"frame{j}"]["sha256"]]
    assert not bad, f"frames {bad} differ from the reference"
    c.close()


def test_cfg3_4k_phase_mode_frame():
    """BASELINE.json configs[2] geometry: one 3840x2160 phase-mode frame of the real reference, from the raw pair."""
    case = "a_3840x2160_phase"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=1)
    rc, frames, dist = c.morph(inp["img1"], inp["img2"], phase=0.5)
    assert rc == 0 and len(frames) == 1
    assert dist == float(G.full(case, "printedMorphDist")[0])
    G.check(case, "frame0", frames[0])
    c.close()


def test_cfg3_4k_sequence_in_flight_equals_single_frames():
    """configs[2] as bench.py runs it: 3840x2160 phase-mode frames of one pair, several in flight, handed to a writer through the pinned ring.
    Every delivered frame must equal the frame rendered on its own (the streaming unsharp kernel, the 128 x 8 warp tiles and the id bytes are
    what a 4K frame takes); the frame at t = 0.5 is the reference's own (fixture a_3840x2160_phase).  The frame-against-itself part is a CONSISTENCY check of
    the in-flight machinery (slots, ring, streams), not parity evidence: the parity of 4K frames is the fixture frame here and test_cfg3_4k_frames_against_the_oracle."""
    case = "a_3840x2160_phase"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=1)
    c.pair_begin(inp["img1"], inp["img2"])
    ts = np.array([k / 121.0 for k in (1, 30, 60, 61, 90, 120)] + [0.5])
    got = []
    c.render_many(ts, chain=False, write=lambda f: got.append(f.copy()))
    assert len(got) == len(ts)
    G.check(case, "frame0", got[-1])
    for t, f in zip(ts[:-1], got[:-1]):
        c.reset()
        assert np.array_equal(c.render(float(t), float(t), chain=False), f), t
    c.close()


def test_cfg3_4k_frames_against_the_oracle():
    """configs[2]: frames 1, 60 and 120 of the 120-frame 3840x2160 phase-mode sequence, as the GPU hands them to a writer, against the oracle's
    frames from the same pair state (three oracle frames at 4K: ~10 s)."""
    import oracle_lib as O
    from poppy_amd import synth
    w, h, n = 3840, 2160, 120
    a, b = synth.gen_pair(w, h, seed=1234)
    c = _ctx(number_of_frames=1)
    c.pair_begin(a, b)
    p1, p2 = c.pair_points()
    g = c.fetch("gabor2")
    ts = np.arange(1, n + 1) / float(n + 1)
    got = {}
    k = [0]

    def write(f):
        if k[0] in (0, 59, 119):
            got[k[0]] = f.copy()
        k[0] += 1
    c.render_many(ts, chain=False, write=write)
    assert k[0] == n and sorted(got) == [0, 59, 119]
    for j, f in got.items():
        want, _ = O.morph_images(a, b, g, p1, p2, float(ts[j]), float(ts[j]), 64)
        assert np.array_equal(want, f), j
    c.close()


def test_writer_and_resident_sequences_alternate_with_a_set_up_between():
    """Phase-mode sequences WITH a writer run on the context's three compute streams (which are also the pair set-up's), sequences whose frames stay
    in HBM on per-slot streams: alternating the two kinds, with a pair set-up in between, must not change a frame (the loaders drain every stream)."""
    case = "a_256x256_phase"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=1)
    ts = np.arange(1, 10) / 10.0
    c.pair_begin(inp["img1"], inp["img2"])
    want = [c.render(float(t), float(t), chain=False).copy() for t in ts]
    for rnd in range(3):
        got = []
        c.render_many(ts, chain=False, write=lambda f: got.append(f.copy()))          # writer: context streams
        assert len(got) == len(ts) and all(np.array_equal(x, y) for x, y in zip(want, got)), rnd
        c.render_many(ts, chain=False)                                                  # resident: slot streams
        assert np.array_equal(c.render(float(ts[-1]), float(ts[-1]), chain=False), want[-1])
        c.pair_begin(inp["img2"], inp["img1"])                                          # another pair on the same streams ...
        other = c.render(0.5, 0.5, chain=False).copy()
        c.pair_begin(inp["img1"], inp["img2"])                                          # ... and back
        assert not np.array_equal(other, want[4])
    c.close()


def test_distance_flag_writes_nothing():
    case = "a_512x512_chain30"
    inp = G.astage_inputs(case)
    c = _ctx(number_of_frames=30)
    rc, frames, dist = c.morph(inp["img1"], inp["img2"], distance=True)
    assert rc == 0 and frames == [] and dist == float(G.full(case, "printedMorphDist")[0])
    assert c.pair_distance() == dist
    c.close()


def test_dissolve_fixture():
    case = "x_dissolve_200x150"
    inp = G.make_inputs.dissolve_inputs(case)
    c = _ctx()
    for k, ph in enumerate(inp["phases"]):
        G.check(case, f"blend{k}", c.dissolve(inp["img1"], inp["img2"], float(ph)), what=f"phase {ph}")
    c.close()


def test_no_match_fallback():
    """A featureless second image leaves no keypoints: poppy_hip_morph writes the frames the reference's fallback branch means
    to write (src/poppy.hpp:125-134, with the phase ARGUMENT, -1 in the default mode) and reports POPPY_E_NOMATCH."""
    import oracle_lib as O
    from poppy_amd import synth
    a, _ = synth.gen_pair(256, 256)
    b = np.full_like(a, 77)
    c = _ctx(number_of_frames=2)
    rc, frames, dist = c.morph(a, b, phase=-1.0)
    assert rc == -5 and len(frames) == 2 and dist is None
    want = O.dissolve(a, b, -1.0)
    assert np.array_equal(frames[0], want) and np.array_equal(frames[1], want)
    c.close()


@pytest.mark.parametrize("case", ["b_640x480_lv4", "b_1920x1080_lv4", "b_320x200_lv1"])
def test_shallow_pyramids(case):
    """--pyramid 4 / 1: the coarsest level is far larger than one workgroup's LDS (was POPPY_E_UNSUPPORTED in round 1)."""
    inp = G.bstage_inputs(case)
    w, h, n, ratios, levels = G.make_inputs.BSTAGE[case]
    c = _ctx(pyramid_levels=levels)
    c.set_debug(True)
    for k, (sr, mr) in enumerate(ratios):
        out, mp = c.morph_images(inp["c1"], inp["c2"], inp["gabor2"], inp["pts1"], inp["pts2"], sr, mr)
        G.check(case, f"f{k}_lbmask", c.fetch("lbmask"))
        G.check(case, f"f{k}_lapBlend", c.fetch("lapBlend"))
        G.check(case, f"f{k}_frame", out)
    c.close()


def test_download_ring_with_two_slots(monkeypatch):
    """POPPY_HIP_SLOTS=2: chained frames, then independent frames with a writer and no reset in between — every delivered frame
    must equal the one rendered with the default slot count."""
    from poppy_amd import capi, synth
    w, h = 320, 192
    c1 = synth.textured_bgr(w, h, 21); c2 = synth.textured_bgr(w, h, 22); g = synth.unit_field(w, h, 11)
    p1, p2 = synth.point_pairs(w, h, 40, seed=5)
    ts = np.linspace(0.1, 0.9, 9)

    def run():
        c = capi.Context(0, number_of_frames=6)
        c.pair_load(c1, c2, g, p1, p2)
        a = []
        c.render_many(np.array([capi.lib().poppy_frame_ratio(j, 6, -1.0) for j in range(6)]), chain=True, write=lambda f: a.append(f.copy()))
        b = []
        c.render_many(ts, chain=False, write=lambda f: b.append(f.copy()))
        c.close()
        return a, b
    ref_a, ref_b = run()
    monkeypatch.setenv("POPPY_HIP_SLOTS", "2")
    got_a, got_b = run()
    assert all(np.array_equal(x, y) for x, y in zip(ref_a, got_a)) and len(got_a) == 6
    assert all(np.array_equal(x, y) for x, y in zip(ref_b, got_b)) and len(got_b) == 9


# ---- multi-GPU entry points, as far as one GPU can exercise them ----------------------------------------------------------
def test_pair_state_export_import_and_single_rank_broadcast():
    """The packed pair state (what travels between GPUs) moved into a second context renders the same frames; a one-rank RCCL
    communicator created through the library broadcasts it in place."""
    import ctypes
    from poppy_amd import capi
    case = "a_256x256_phase"
    inp = G.astage_inputs(case)
    h, w = inp["img1"].shape[:2]
    a = _ctx(number_of_frames=1)
    a.pair_begin(inp["img1"], inp["img2"])
    n = capi.pair_state_bytes(w, h)
    hip = ctypes.CDLL("libamdhip64.so")             # the HIP runtime the library already runs on (no torch in this process)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), ctypes.c_size_t(n)) == 0
    a.pair_export_device(buf, n)
    b = _ctx(number_of_frames=1)
    b.pair_import_device(buf, n, w, h)
    hip.hipFree(buf)
    pa, pb = a.pair_points(), b.pair_points()
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
    fb = b.morph_frames(0.5)
    G.check(case, "frame0", fb[0])
    # one-rank communicator: init, broadcast (root = the only rank), max
    a.comm_init(0, 1, capi.comm_id())
    a.pair_broadcast(0, w, h)
    assert a.comm_max(3.25) == 3.25
    G.check(case, "frame0", a.morph_frames(0.5)[0])
    a.comm_free()
    a.close(); b.close()


def test_morph_sharded_one_device_equals_phase_frames():
    """poppy_hip_morph_sharded on one device: frame j = morph(.., phase = j / total) with number_of_frames = 1; frame 0 = image 1."""
    from poppy_amd import capi
    case = "a_1920x1080_chain60"
    inp = G.astage_inputs(case)
    frames = capi.morph_sharded([0], inp["img1"], inp["img2"], 4)
    assert len(frames) == 4
    assert np.array_equal(frames[0], inp["img1"])
    G.check(case, "phase0_frame", frames[1])          # t = 1/4
    G.check(case, "phase1_frame", frames[2])          # t = 2/4


def test_cfg4_480_frame_job_on_one_device():
    """BASELINE.json configs[3] at full size on the one device a test box has: poppy_hip_morph_sharded([0], 1080p pair, 480) — the whole job a node would
    split eight ways.  Frame 0 is image 1 (the phase == 0 short-circuit, src/poppy.hpp:54-70); frames 120 and 240 (t = 0.25, 0.5) are the REAL reference's
    phase-mode frames (fixture a_1920x1080_chain60); four more sampled frames equal the oracle's from the same pair state; and all 480 equal, by sha256,
    what poppy_hip_render_phases makes of the same t_j on an ordinary context (property: a shard's frames are the single-GPU frames)."""
    import hashlib
    import oracle_lib as O
    from poppy_amd import capi
    case = "a_1920x1080_chain60"
    inp = G.astage_inputs(case)
    total = 480
    sampled = {0: None, 1: None, 97: None, 120: None, 240: None, 311: None, 479: None}

    def keep(idx, f):
        if idx in sampled:
            sampled[idx] = f.copy()
        return hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest()
    shas = capi.morph_sharded([0], inp["img1"], inp["img2"], total, reduce=keep)
    assert len(shas) == total
    assert np.array_equal(sampled[0], inp["img1"])
    G.check(case, "phase0_frame", sampled[120])
    G.check(case, "phase1_frame", sampled[240])
    c = _ctx(number_of_frames=1)
    c.pair_begin(inp["img1"], inp["img2"])
    p1, p2 = c.pair_points()
    g = c.fetch("gabor2")
    for j in (1, 97, 311, 479):
        t = j / float(total)
        want, _ = O.morph_images(inp["img1"], inp["img2"], g, p1, p2, t, t, 64)
        assert np.array_equal(want, sampled[j]), j
    got = []
    c.render_many(np.arange(1, total) / float(total), chain=False, write=lambda f: got.append(hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest()))
    c.close()
    assert got == shas[1:], [j + 1 for j in range(total - 1) if got[j] != shas[j + 1]][:8]
    # the whole job's digest (a checksum of checksums), for the record of what ran
    print("cfg4 480-frame job sha256:", hashlib.sha256("".join(shas).encode()).hexdigest())


def test_cfg5_pooled_1080p_pairs_on_one_device():
    """BASELINE.json configs[4] at full size on one device: poppy_hip_morph_pairs([0], 2 x the 1080p pair x 60 chained frames, contexts_per_device = 2) — what a
    GPU of the node renders of the 64-pair batch, two pairs side by side.  Every frame of both pairs against the real reference's (fixture a_1920x1080_chain60, sha256)."""
    import hashlib
    from poppy_amd import capi
    case = "a_1920x1080_chain60"
    inp = G.astage_inputs(case)
    out = capi.morph_pairs([0], [(inp["img1"], inp["img2"])] * 2, contexts_per_device=2, number_of_frames=60,
                           reduce=lambda p, j, f: hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest())
    assert sorted(out) == [0, 1]
    for p in out:
        assert len(out[p]) == 60
        bad = [j for j, h in enumerate(out[p]) if h != G.entries(case)[f"frame{j}"]["sha256"]]
        assert not bad, f"pair {p}: frames {bad} differ from the reference"


def test_morph_pairs_one_device():
    """poppy_hip_morph_pairs: three pairs over two contexts of one GPU; each pair's frames equal the single-context run."""
    from poppy_amd import capi
    cases = ["a_256x256_chain", "a_256x256_chain", "a_256x256_chain"]
    inp = G.astage_inputs(cases[0])
    n = int(inp["cfg"][0])
    out = capi.morph_pairs([0], [(inp["img1"], inp["img2"])] * 3, contexts_per_device=2, number_of_frames=n)
    assert sorted(out) == [0, 1, 2]
    for p in out:
        assert len(out[p]) == n
        for j, f in enumerate(out[p]):
            G.check(cases[p], f"frame{j}", f)


def test_morph_pairs_on_three_contexts():
    """A pool of three contexts per device runs a pair set-up's two image chains one after the other (poppy_hip_set_setup_chains, set by poppy_hip_pool_create from
    three contexts on): four pairs over three contexts, every frame of every pair as the reference's."""
    from poppy_amd import capi
    case = "a_256x256_chain"
    inp = G.astage_inputs(case)
    n = int(inp["cfg"][0])
    out = capi.morph_pairs([0], [(inp["img1"], inp["img2"])] * 4, contexts_per_device=3, number_of_frames=n)
    assert sorted(out) == [0, 1, 2, 3]
    for p in out:
        assert len(out[p]) == n
        for j, f in enumerate(out[p]):
            G.check(case, f"frame{j}", f)


def test_pool_batches_queued_without_waiting():
    """poppy_hip_pool_submit_pairs / poppy_hip_pool_wait: three batches (2 + 3 + 1 pairs) queued back to back on a pool of three contexts, then one wait — the batches
    overlap (a batch's last pair renders beside the next batch's set-ups); every frame of every pair as the real reference's; a second round on the same pool after
    the wait; then a batch whose pair source fails: the wait reports it, and the pool still renders afterwards."""
    import threading
    from poppy_amd import capi
    case = "a_256x256_chain"
    inp = G.astage_inputs(case)
    n = int(inp["cfg"][0])
    pool = capi.Pool([0], contexts_per_device=3, number_of_frames=n)
    lock = threading.Lock()
    try:
        for rnd in range(2):
            got = {}
            def writer(b):
                def w(pair, j, view):
                    f = view.copy()
                    with lock:
                        got.setdefault((b, pair), {})[j] = f
                return w
            for b, k in enumerate((2, 3, 1)):
                pool.submit_pairs([(inp["img1"], inp["img2"])] * k, writer(b))
            pool.wait()
            assert sorted(got) == [(0, 0), (0, 1), (1, 0), (1, 1), (1, 2), (2, 0)], sorted(got)
            for key, frames in got.items():
                assert sorted(frames) == list(range(n)), (key, sorted(frames))
                for j in range(n):
                    G.check(case, f"frame{j}", frames[j])
        # a failing pair source: reported by the wait, nothing hangs, the pool is usable afterwards
        import ctypes as C
        bad = capi.PAIR_SOURCE_CB(lambda user, p, device, pa, sa, pb, sb: 1)
        rc = capi.lib().poppy_hip_pool_submit_pairs(pool.h, 4, 256, 256, -1.0, 0, C.cast(bad, C.c_void_p), None, None)
        assert rc == 0
        with pytest.raises(capi.PoppyError, match="pair source"):
            pool.wait()
        got = {}
        pool.submit_pairs([(inp["img1"], inp["img2"])], lambda pair, j, view: got.setdefault(j, view.copy()))
        pool.wait()
        assert sorted(got) == list(range(n))
        G.check(case, f"frame{n - 1}", got[n - 1])
    finally:
        pool.close()


def test_tuned_pool_and_communicator_info():
    """poppy_hip_pool_create_tuned makes candidate pools, times the built-in calibration batch on each and hands out one of them, which then renders device-resident
    pairs like any pool; poppy_hip_comm_info without a communicator reports -1 for what RCCL would say, and the world of one after comm_init."""
    import ctypes as C
    from poppy_amd import capi
    inp = G.astage_inputs("a_256x256_chain")
    n = int(inp["cfg"][0])
    h, w = inp["img1"].shape[:2]
    pool = capi.Pool([0], contexts_per_device=2, tuned_for=(w, h), max_candidates=2, number_of_frames=n)
    assert 1 <= len(pool.candidates_ms) <= 2 and all(ms > 0 for ms in pool.candidates_ms) and 0 <= pool.kept < len(pool.candidates_ms)
    hip = C.CDLL("libamdhip64.so")
    ptrs = []
    for img in (inp["img1"], inp["img2"]):
        a = np.ascontiguousarray(img)
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(d, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
        ptrs.append(d.value)
    assert pool.morph_pairs_device_counted([tuple(ptrs)] * 3, w, h, -1.0) == 3 * n
    pool.close()
    for p in ptrs:
        hip.hipFree(C.c_void_p(p))
    ctx = capi.Context(0, number_of_frames=1)
    assert ctx.comm_info()[2:] == (-1, -1)
    ctx.comm_init(0, 1, capi.comm_id())
    assert ctx.comm_info() == (0, 1, 0, 1)
    ctx.comm_free()
    ctx.close()


def test_pool_refuses_several_pairs_under_auto_align():
    """With --autoalign the reference's pairs form a chain (src/poppy.cpp:326: img1 = corrected2.clone(), the ALIGNED image): the pool, which
    hands pairs out concurrently, refuses more than one of them instead of silently rendering a different sequence."""
    from poppy_amd import capi
    inp = G.astage_inputs("a_256x256_chain")
    n = int(inp["cfg"][0])
    with pytest.raises(capi.PoppyError, match="auto_align"):
        capi.morph_pairs([0], [(inp["img1"], inp["img2"])] * 2, contexts_per_device=2, number_of_frames=n, enable_auto_align=1)
    out = capi.morph_pairs([0], [(inp["img1"], inp["img2"])], contexts_per_device=2, number_of_frames=n, enable_auto_align=1)    # one pair: fine
    assert len(out[0]) == n


def test_frame_wait_orders_a_caller_stream_behind_a_phase_mode_frame():
    """Phase-mode frames run on per-slot streams: poppy_hip_frame_wait makes a caller's stream (here: the context's own) wait for the
    last frame, so that device work queued there reads the finished image (the hand-off contract of include/poppy_hip.h)."""
    import ctypes as C
    from poppy_amd import capi
    inp = G.astage_inputs("a_256x256_phase")
    ctx = capi.Context(0, number_of_frames=1)
    ctx.pair_begin(inp["img1"], inp["img2"])
    want = ctx.render(0.5, 0.5, chain=False)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    got = np.zeros_like(want)
    for _ in range(20):
        ctx.render_many(np.array([0.25, 0.5]), chain=False)                 # two frames in flight, nothing downloaded, no sync
        assert ctx.frame_stream_ptr() != ctx.stream_ptr()
        ctx.frame_wait(ctx.stream_ptr())
        assert hip.hipMemcpyAsync(got.ctypes.data, ctx.frame_device_ptr(), got.nbytes, 2, ctx.stream_ptr()) == 0    # 2 = device to host
        assert hip.hipStreamSynchronize(ctx.stream_ptr()) == 0
        assert (got == want).all()
    ctx.close()


@pytest.mark.parametrize("n_ctx,root", [(2, 0), (3, 0), (3, 1), (4, 2)])
def test_sharded_pair_setup_between_contexts_equals_the_one_gpu_setup(n_ctx, root):
    """The pair set-up spread over ranks (comm.cpp: setup_sharded — image 1 on rank root, image 2 on root + 1, the mask field on root + 2,
    five small exchanges) run between contexts of this process on one GPU: every context ends up with the pair state of the one-GPU set-up —
    same point sets, same nfeatures, and the reference's frames."""
    import ctypes as C
    from poppy_amd import capi
    case = "a_256x256_phase"
    inp = G.astage_inputs(case)
    h, w = inp["img1"].shape[:2]
    ref = _ctx(number_of_frames=1)
    nf, _ = ref.pair_begin(inp["img1"], inp["img2"])
    want_pts = ref.pair_points()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    d = [C.c_void_p(), C.c_void_p()]
    for k, img in enumerate((inp["img1"], inp["img2"])):
        img = np.ascontiguousarray(img)
        assert hip.hipMalloc(C.byref(d[k]), C.c_size_t(img.nbytes)) == 0
        assert hip.hipMemcpy(d[k], img.ctypes.data, img.nbytes, 1) == 0          # 1 = host to device
    ctxs = [_ctx(number_of_frames=1) for _ in range(n_ctx)]
    for rep in range(2):                                                           # twice: the second run reuses every buffer
        capi.pair_begin_sharded_local(ctxs, d[0], d[1], w, h, root)
        for c in ctxs:
            p = c.pair_points()
            assert np.array_equal(p[0], want_pts[0]) and np.array_equal(p[1], want_pts[1])
            G.check(case, "frame0", c.morph_frames(0.5)[0])
    assert ctxs[root].pair_begin_info()[0] == nf
    for k in range(2):
        hip.hipFree(d[k])
    for c in ctxs:
        c.close()
    ref.close()


def test_bench_sharded_path_on_a_world_of_one():
    """bench.py's N > 1 code path (library communicator, the two forms of the pair set-up timed and compared, frame shares, cfg5 pairs) on a
    world of ONE rank over RCCL (POPPY_BENCH_SHARDED_SELFTEST): the only way this box can execute it.  The self-test sets POPPY_HIP_SHARD_WORLD1, so
    the sharded form IS the protocol (all three roles on rank 0, every broadcast and reduction an RCCL call of the library's dlopen'ed copy, with
    torch's own RCCL initialised in the same process: the two-copies case), not the world-of-one shortcut.  The forms must leave the same point lists."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POPPY_BENCH_SHARDED_SELFTEST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--total-frames", "64", "--pairs-per-gpu", "2"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    f = d["setup_forms"]
    assert f["same_point_lists_on_every_rank"] is True and f["used"] in ("sharded", "rank 0 + broadcast")
    assert f["sharded_protocol_runs_on_rank0"] >= 4          # 1 + 3 timed runs of the protocol itself
    assert d["cfg5_pairs"]["value"] > 0


@pytest.mark.parametrize("case,pads", [("a_256x256_phase", (13, 13)), ("a_256x256_chain", (13, 40)), ("a_639x480_numbers", (13, 1))])
def test_main_entry_takes_images_with_padded_rows(case, pads):
    """poppy_hip_morph on images whose row stride is not 3 * width — the reference hands poppy::morph ROI cv::Mats of the union canvas (src/poppy.cpp:234-239:
    step > cols * 3) — with a different padding per image, the padding bytes filled with a pattern: points, distance and every frame against the reference's
    fixture of the same pair handed over tight (an odd-width pair among them: the upload then takes the 2-D copy)."""
    inp = G.astage_inputs(case)
    n = int(inp["cfg"][0]); phase = float(inp["cfg"][1])
    c = _ctx(number_of_frames=n)
    rc, frames, dist = c.morph(inp["img1"], inp["img2"], phase=phase, row_pad=pads)
    assert rc == 0 and len(frames) == (n if phase < 0 else 1)
    if "printedMorphDist" in G.entries(case):
        assert dist == float(G.full(case, "printedMorphDist")[0])
    p1, p2 = c.pair_points()
    G.check(case, "prepared1", p1); G.check(case, "prepared2", p2)
    for j, f in enumerate(frames):
        assert G.sha(f) == G.entries(case)[f"frame{j}"]["sha256"], f"frame {j}"
    c.close()
