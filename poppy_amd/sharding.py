"""Frame-range sharding of one phase-mode morph across the ranks of a node (BASELINE.json configs[3]).

The job is ONE morph of `frames_per_rank * world` frames with phase t_j = j / total; frame j equals the reference call
morph(img1, img2, ..., phase = t_j) with number_of_frames = 1 (src/poppy.hpp:177-210,234-235; t_0 = 0 takes the phase == 0
short-circuit of :54-62 and is a copy of image 1: poppy_hip_render_phases), so frames are
independent and rank r renders the contiguous range [r * frames_per_rank, (r + 1) * frames_per_rank).  The only
exchange is one broadcast of the pair (sources, mask field, point sets) from rank 0; there is no data-path
collective afterwards.  Default (chained) mode cannot be sharded: frame j warps frame j-1 (src/poppy.hpp:217).

Everything here is backend agnostic: bench.py runs it over "nccl" (RCCL on xGMI), the CPU tests over "gloo".
"""
import numpy as np


def frame_range(rank, world, frames_per_rank):
    """Global frame indices rendered by `rank`."""
    if not (0 <= rank < world) or frames_per_rank < 0:
        raise ValueError("bad rank / world / frames_per_rank")
    return range(rank * frames_per_rank, (rank + 1) * frames_per_rank)


def phase_schedule(rank, world, frames_per_rank):
    """shape = mask ratio of every frame of `rank`: t_j = j / (frames_per_rank * world), float64 as the reference's."""
    total = float(frames_per_rank * world)
    return np.array([j / total for j in frame_range(rank, world, frames_per_rank)], dtype=np.float64)


def frame_share(rank, world, total_frames):
    """Global frame indices of `rank` when ONE total_frames-frame morph is split over `world` ranks (strong scaling: the job is fixed):
    the rank-th contiguous share, [total * rank / world, total * (rank + 1) / world) — what poppy_hip_morph_sharded gives device k."""
    if not (0 <= rank < world) or total_frames < 0:
        raise ValueError("bad rank / world / total_frames")
    return range(total_frames * rank // world, total_frames * (rank + 1) // world)


def phase_share(rank, world, total_frames):
    """shape = mask ratio of every frame of `rank`'s share of the fixed job: t_j = j / total_frames (float64 as the reference's)."""
    return np.array([j / float(total_frames) for j in frame_share(rank, world, total_frames)], dtype=np.float64)


def pair_range(rank, world, n_pairs):
    """Pairs rendered by `rank` when n_pairs independent pairs are split over the ranks (BASELINE.json configs[4]: 64 pairs over 8
    GPUs): contiguous, disjoint, complete; the first n_pairs % world ranks take one more."""
    if not (0 <= rank < world) or n_pairs < 0:
        raise ValueError("bad rank / world / n_pairs")
    base, extra = divmod(n_pairs, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def pair_tensors(torch, device, w, h, n_points, host_inputs=None):
    """The four tensors that make up a pair on `device`: sources a, b (u8 HxWx3), mask field g (f32 HxWx3) and the
    two point sets stacked (f32 2xNx2).  Rank 0 passes `host_inputs` = (a, b, g, p1, p2) numpy arrays; the other
    ranks get empty tensors of the same shapes to receive the broadcast into."""
    if host_inputs is not None:
        a, b, g, p1, p2 = host_inputs
        return [torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), torch.from_numpy(g).to(device),
                torch.from_numpy(np.stack([p1, p2])).to(device)]
    return [torch.empty((h, w, 3), dtype=torch.uint8, device=device), torch.empty((h, w, 3), dtype=torch.uint8, device=device),
            torch.empty((h, w, 3), dtype=torch.float32, device=device), torch.empty((2, n_points, 2), dtype=torch.float32, device=device)]


def broadcast_pair(dist, tensors, src=0):
    """One broadcast per tensor from `src`; 12.4 MB + 24.9 MB + a few KB at 1080p."""
    for t in tensors:
        dist.broadcast(t, src=src)
    return tensors


class PairLink:
    """Moves the resident pair of rank `root` into every rank's context, once per step.

    Preferred transport: the library's own RCCL communicator (poppy_hip_comm_init + poppy_hip_pair_broadcast: ONE ncclBroadcast of
    the packed pair state straight out of the context's arena).  The 128-byte unique id is the only thing that goes through
    torch.distributed.  If that communicator cannot be created (or the job runs on the gloo rehearsal path), the same packed
    state is exported into a torch tensor, broadcast by torch.distributed and imported — still one broadcast, one more copy."""

    def __init__(self, torch, dist, capi, ctx, rank, world, w, h, device, use_library=True):
        self.torch, self.dist, self.capi, self.ctx, self.rank, self.world, self.w, self.h, self.device = torch, dist, capi, ctx, rank, world, w, h, device
        self.nbytes = capi.pair_state_bytes(w, h)
        self.library = False
        self._buf = None
        ok = 1
        if use_library and device.type == "cuda":
            ident = torch.zeros(128, dtype=torch.uint8, device=device)
            if rank == 0:
                try:
                    ident = torch.frombuffer(bytearray(capi.comm_id()), dtype=torch.uint8).to(device)
                except Exception:
                    ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.broadcast(flag, src=0)
            if int(flag.item()):
                dist.broadcast(ident, src=0)
                try:
                    ctx.comm_init(rank, world, bytes(ident.cpu().numpy().tobytes()))
                except Exception:
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                self.library = bool(int(flag.item()))
                if not self.library:
                    ctx.comm_free()
        self.how = ("ncclBroadcast of the packed pair state through the library's own RCCL communicator" if self.library
                    else "torch.distributed broadcast of the packed pair state (export / import through a device buffer)")

    def broadcast(self, root=0):
        if self.library:
            self.ctx.pair_broadcast(root, self.w, self.h)
            return
        torch = self.torch
        if self._buf is None:
            self._buf = torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)
        if self.device.type == "cuda":
            if self.rank == root:
                self.ctx.pair_export_device(self._buf.data_ptr(), self.nbytes)
            self.dist.broadcast(self._buf, src=root)
            torch.cuda.synchronize()
            if self.rank != root:
                self.ctx.pair_import_device(self._buf.data_ptr(), self.nbytes, self.w, self.h)
        else:                                   # gloo rehearsal: staged through the host
            stage = torch.empty(self.nbytes, dtype=torch.uint8, device="cuda") if not hasattr(self, "_stage") else self._stage
            self._stage = stage
            if self.rank == root:
                self.ctx.pair_export_device(stage.data_ptr(), self.nbytes)
                self._buf.copy_(stage.cpu())
            self.dist.broadcast(self._buf, src=root)
            if self.rank != root:
                stage.copy_(self._buf)
                torch.cuda.synchronize()
                self.ctx.pair_import_device(stage.data_ptr(), self.nbytes, self.w, self.h)

    def max_time(self, seconds):
        """Slowest rank's step time."""
        if self.library:
            return self.ctx.comm_max(seconds)
        return max_over_ranks(self.torch, self.dist, seconds, self.device)


def max_over_ranks(torch, dist, seconds, device):
    """Step time of the job = the slowest rank's."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
