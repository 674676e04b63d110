"""Frame -> GPU assignment for a stream processed by several replicas.

Frames are independent units (for history-free configurations), so a stream shards by frame with no
data-path collective: rank r of W processes frames r, r + W, r + 2W, ... (round robin keeps every
GPU's working set in stream order) or a contiguous block.  torch.distributed is used only to agree
on the elapsed time (barrier + max-reduce), exactly as bench.py does.
"""
from __future__ import annotations


def frames_for_rank(n_frames: int, rank: int, world: int, mode: str = "round_robin") -> list[int]:
    """Indices of the frames rank `rank` of `world` processes."""
    if not 0 <= rank < world:
        raise ValueError("rank outside 0..world-1")
    if mode == "round_robin":
        return list(range(rank, n_frames, world))
    if mode == "block":
        per, extra = divmod(n_frames, world)
        start = rank * per + min(rank, extra)
        return list(range(start, start + per + (1 if rank < extra else 0)))
    raise ValueError(mode)


def owner_of(frame: int, n_frames: int, world: int, mode: str = "round_robin") -> int:
    """Inverse of frames_for_rank."""
    if mode == "round_robin":
        return frame % world
    per, extra = divmod(n_frames, world)
    edge = extra * (per + 1)
    return frame // (per + 1) if frame < edge else extra + (frame - edge) // max(per, 1)


def max_elapsed(elapsed: float, device=None) -> float:
    """Max over ranks of a locally measured time (the bench's only collective besides the barrier)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
