"""Batch sharding and the one exchange step of data-parallel training.

Replaces torch.nn.parallel.DataParallel at /root/reference/btsbot/train.py:238-240: instead of
re-broadcasting every parameter and scattering inputs from device 0 each step, every rank keeps a
persistent replica and its own shard of alerts; the only collective is ONE all-reduce of the flat
gradient arena per step (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n alerts for `rank`; the first n % world ranks get one extra."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """Sum `flat` over ranks in place.  The local gradients are already scaled by 1/B_global
    (btsbot_bce_fwd_bwd's n_global), so the SUM is the gradient of the global-batch mean loss."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Make every replica start from rank `src`'s parameters (what DataParallel's per-step
    broadcast guaranteed implicitly)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat
