"""Frame-range sharding of one phase-mode morph across the ranks of a node (BASELINE.json configs[3]).

The job is ONE morph of `frames_per_rank * world` frames with phase t_j = j / total; frame j equals the reference call
morph(img1, img2, ..., phase = t_j) with number_of_frames = 1 (src/poppy.hpp:177-210,234-235), so frames are
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


def max_over_ranks(torch, dist, seconds, device):
    """Step time of the job = the slowest rank's."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
