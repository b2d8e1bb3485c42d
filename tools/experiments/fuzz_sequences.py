"""Differential fuzz of the SEQUENCE machinery (round 6: frames prepared ahead, plans started by the pair loader, frame copies on event-free streams, the pool's set-up
gate): random image pairs and frame counts; the frames of poppy_hip_morph (whole sequence, writer attached, chained) against the same frames rendered ONE BY ONE with
poppy_hip_render on a second context (no prefetch, no writer ring) — and, for every fifth case, against the pool (poppy_hip_morph_pairs, three contexts) and against a
persistent pool fed two QUEUED batches (poppy_hip_pool_submit_pairs twice, one poppy_hip_pool_wait; round 6, last day).  The single-frame
path is what fuzz_frames.py holds to the oracle.   python tools/experiments/fuzz_sequences.py [cases] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from poppy_amd import capi, synth
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; frames_total = 0; nomatch = 0; t0 = time.time()
ctxs = {}
pools = {}
import threading
for i in range(cases):
    w = int(rng.integers(96, 700)); h = int(rng.integers(80, 480))
    if rng.random() < 0.6: w &= ~3
    n = int(rng.integers(2, 14))
    seed = int(rng.integers(1, 10**6))
    a, b = synth.gen_pair(w, h, seed=seed)
    key = n
    if key not in ctxs: ctxs[key] = (capi.Context(0, number_of_frames=n), capi.Context(0, number_of_frames=n))
    A, B = ctxs[key]
    rc, frames, _ = A.morph(a, b)
    if rc == -5: nomatch += 1; continue
    assert len(frames) == n
    B.pair_begin(a, b)
    for j in range(n):
        r = capi.lib().poppy_frame_ratio(j, n, -1.0)
        f = B.render(r, r, chain=True)
        if not np.array_equal(f, frames[j]):
            print(f"case {i} {w}x{h} n={n} seed={seed}: frame {j} of the sequence differs from the frame rendered alone ({int((f != frames[j]).sum())} bytes)"); bad += 1; break
    frames_total += n
    if i % 5 == 0:
        out = capi.morph_pairs([0], [(a, b)] * 4, contexts_per_device=3, number_of_frames=n)
        for p in out:
            if len(out[p]) != n or any(not np.array_equal(x, y) for x, y in zip(out[p], frames)):
                print(f"case {i} {w}x{h} n={n} seed={seed}: pair {p} of the pool differs"); bad += 1; break
        if n not in pools: pools[n] = capi.Pool([0], contexts_per_device=3, number_of_frames=n)
        got, lock = {}, threading.Lock()
        def writer(bi):
            def wr(p, j, view):
                f = view.copy()
                with lock: got.setdefault((bi, p), {})[j] = f
            return wr
        pools[n].submit_pairs([(a, b)] * 3, writer(0)); pools[n].submit_pairs([(a, b)] * 2, writer(1))
        pools[n].wait()
        if sorted(got) != [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)] or any(sorted(v) != list(range(n)) or any(not np.array_equal(v[j], frames[j]) for j in range(n)) for v in got.values()):
            print(f"case {i} {w}x{h} n={n} seed={seed}: the queued batches differ"); bad += 1
print(f"{cases} cases ({nomatch} without matches), {frames_total} frames, {bad} mismatches, {time.time() - t0:.0f} s")
