"""Multi-GPU layer: shard faces over ranks, collate poses with ONE all-gather.

The reference is single-process; every face is independent
(NLML_HPE_Model_Builder.py:55-68,104-126 have no cross-row op), so the path shards with no
data-path exchange: rank r owns the contiguous row block [r*ceil(N/R), (r+1)*ceil(N/R)), weights are
replicated (9.5 MB), and the only collective is the all-gather of the f32[rows,3] poses
(BASELINE.json north_star).  On ROCm the "nccl" backend is RCCL over xGMI; the message is 12 B per
face (786 KB per rank at 65,536 faces) -- latency-bound, so it is issued asynchronously and the
next batch's compute overlaps it.  "gloo" works the same way on CPU tensors (used by the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> tuple:
    """Contiguous row block of `rank`: (start, stop, rows_per_rank) with rows_per_rank = ceil(n/world)."""
    per = (n + world - 1) // world
    start = min(rank * per, n)
    return start, min(start + per, n), per


class PoseGatherer:
    """Double-buffered asynchronous all-gather of per-rank pose blocks f32[rows,3]."""

    def __init__(self, rows_per_rank: int, world: int, device, dtype=torch.float32, depth: int = 2, group=None):
        self.rows, self.world, self.group, self.depth = rows_per_rank, world, group, depth
        self.gathered = [torch.empty((world * rows_per_rank, 3), dtype=dtype, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.src = [None] * depth      # keeps the source block alive while its collective is in flight
        self.i = 0

    def submit(self, pose: torch.Tensor) -> int:
        if tuple(pose.shape) != (self.rows, 3):
            raise ValueError(f"pose block must be [{self.rows},3], got {tuple(pose.shape)}")
        slot = self.i % self.depth
        if self.work[slot] is not None:
            self.work[slot].wait()
        self.src[slot] = pose
        self.work[slot] = dist.all_gather_into_tensor(self.gathered[slot], pose.contiguous(), group=self.group, async_op=True)
        self.i += 1
        return slot

    def drain(self) -> torch.Tensor | None:
        for s in range(self.depth):
            if self.work[s] is not None:
                self.work[s].wait()
                self.work[s] = None
        return self.gathered[(self.i - 1) % self.depth] if self.i else None


def gather_poses(local_pose: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All ranks call with their block (rank r holds rows shard_bounds(n_total, R, r)); returns f32[n_total,3]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    start, stop, per = shard_bounds(n_total, world, rank)
    if local_pose.shape[0] != stop - start:
        raise ValueError(f"rank {rank} must hold {stop - start} rows, got {local_pose.shape[0]}")
    block = local_pose
    if stop - start < per:  # pad the short (last) blocks so every rank sends the same count
        pad = torch.zeros((per - (stop - start), 3), dtype=local_pose.dtype, device=local_pose.device)
        block = torch.cat([local_pose, pad], dim=0)
    out = torch.empty((world * per, 3), dtype=local_pose.dtype, device=local_pose.device)
    dist.all_gather_into_tensor(out, block.contiguous(), group=group)
    return out[:n_total]
