"""N > 1 path of bench.py on CPU: two processes over gloo (the GPU run uses the same code over nccl = RCCL).

Covers poppy_amd/sharding.py: the pair broadcast, the frame-range partition of one phase-mode morph, the
max-over-ranks step time, and that a rank's frames are planned exactly as the unsharded job would plan the same
global frame indices (host planner through the C ABI, no GPU needed).
"""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from poppy_amd import sharding  # noqa: E402

W, H, NPTS, FPR = 160, 96, 40, 6


def _inputs():
    from poppy_amd import synth
    a, b = synth.gen_pair(W, H)
    g = synth.unit_field(W, H, 11)
    p1, p2 = synth.point_pairs(W, H, NPTS, seed=5, dup=0, oob=0)
    return a, b, g, p1, p2


def _digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _plan_digest(p1, p2, shape):
    from poppy_amd import capi
    plan = capi.plan_frame(W, H, p1, p2, float(shape))
    return _digest(plan["tri_xy"], plan["inv1"], plan["inv2"], plan["morphed"])


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        host = _inputs() if rank == 0 else None
        n_pts = NPTS + 4
        tensors = sharding.pair_tensors(torch, dev, W, H, n_pts, host)
        sharding.broadcast_pair(dist, tensors, src=0)
        ta, tb, tg, tp = [t.numpy() for t in tensors]
        shapes = sharding.phase_schedule(rank, world, FPR)
        plans = [_plan_digest(np.ascontiguousarray(tp[0]), np.ascontiguousarray(tp[1]), s) for s in shapes]
        tmax = sharding.max_over_ranks(torch, dist, 1.0 + rank, dev)
        out.put((rank, _digest(ta, tb, tg, tp), list(sharding.frame_range(rank, world, FPR)), shapes.tolist(), plans, tmax))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_over_gloo(world):
    """Two and three ranks (three = the number of roles of the sharded pair set-up: image 1, image 2, the mask field)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    a, b, g, p1, p2 = _inputs()
    want_pair = _digest(a, b, g, np.stack([p1, p2]))
    frames, shapes, plans = [], [], []
    for rank, pair, fr, sh, pl, tmax in results:
        assert pair == want_pair                      # every rank holds rank 0's pair after the broadcast
        assert tmax == float(world)                   # slowest rank's time (1 + rank)
        frames += fr; shapes += sh; plans += pl
    total = FPR * world
    assert frames == list(range(total))               # contiguous, disjoint, complete
    assert shapes == [j / float(total) for j in range(total)]
    # the sharded job plans every frame exactly as the unsharded 1-rank job does
    assert plans == [_plan_digest(p1, p2, s) for s in sharding.phase_schedule(0, 1, total)]


def test_pair_partition():
    """configs[4]: independent pairs over ranks — contiguous, disjoint, complete, balanced to within one pair."""
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 64, 65):
            seen, sizes = [], []
            for r in range(world):
                pr = list(sharding.pair_range(r, world, n))
                seen += pr; sizes.append(len(pr))
            assert seen == list(range(n))
            assert max(sizes) - min(sizes) <= 1
    assert list(sharding.pair_range(3, 8, 64)) == list(range(24, 32))
    with pytest.raises(ValueError):
        sharding.pair_range(8, 8, 64)


def test_frame_zero_of_the_sharded_job_is_the_phase_zero_shortcut():
    """Global frame 0 has t = 0: only rank 0 holds it, and the scheduler the library applies to a phase-mode call with phase 0 < t < 1
    never yields it (the reference short-circuits phase == 0 before any rendering, src/poppy.hpp:54-62)."""
    for world in (1, 2, 8):
        for r in range(world):
            ts = sharding.phase_schedule(r, world, 60)
            assert (ts[0] == 0.0) == (r == 0)
            assert (ts[1:] > 0).all() and (ts < 1).all()


def test_partition_properties():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += list(sharding.frame_range(r, world, 60))
        assert seen == list(range(60 * world))
        all_t = np.concatenate([sharding.phase_schedule(r, world, 60) for r in range(world)])
        assert np.array_equal(all_t, np.arange(60 * world) / float(60 * world))
    with pytest.raises(ValueError):
        sharding.frame_range(2, 2, 60)


def test_fixed_job_shares():
    """Strong scaling: ONE total-frame morph split over N ranks (bench.py --gpus N, poppy_hip_morph_sharded): contiguous, disjoint, complete
    shares of [0, total), every frame's phase is j / total whatever N is, and the shares agree with the weak-scaling schedule when total = 60 N."""
    for total in (480, 481, 7, 1):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                fr = list(sharding.frame_share(r, world, total))
                assert fr == list(range(total * r // world, total * (r + 1) // world))
                ts = sharding.phase_share(r, world, total)
                assert np.array_equal(ts, np.array(fr, dtype=np.float64) / float(total))
                seen += fr
            assert seen == list(range(total))
    for world in (1, 2, 4, 8):
        for r in range(world):
            assert np.array_equal(sharding.phase_share(r, world, 60 * world), sharding.phase_schedule(r, world, 60))
    assert len(sharding.frame_share(7, 8, 480)) == 60
    with pytest.raises(ValueError):
        sharding.frame_share(8, 8, 480)
