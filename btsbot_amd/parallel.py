"""Batch sharding and the one exchange step of data-parallel training.

Replaces torch.nn.parallel.DataParallel at /root/reference/btsbot/train.py:238-240: instead of
re-broadcasting every parameter and scattering inputs from device 0 each step, every rank keeps a
persistent replica (made equal ONCE, ``broadcast_`` at ``Trainer`` construction) and its own shard of
alerts; the only collective of a step is the all-reduce (SUM) of the trainable part of the flat gradient
arena -- RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.  The arena is exchanged in a few
buckets in the order ``btsbot_backward`` finishes them (fusion head / metadata branch / last image stage
first, stem last), each on a side stream as soon as the library signals it, so the collectives run under
the rest of the backward pass.

Everything here is host logic over plain tensors (no HIP call): the CPU tests drive the same code with
gloo at world sizes 2 and 3.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n alerts for `rank`; the first n % world ranks get one extra."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world(group=None) -> int:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def trainable_ranges(slots: Iterable[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """Merged arena ranges [lo, hi) of the tensors that are trained.  `slots` = (offset, numel) of every
    tensor with requires_grad; tensors are 4-float aligned in the arena (the padding carries zero gradient),
    so neighbours merge; frozen tensors and BatchNorm buffers leave gaps."""
    out: List[List[int]] = []
    for lo, n in sorted(slots):
        hi = (lo + n + 3) // 4 * 4
        if out and lo <= out[-1][1]:
            out[-1][1] = max(hi, out[-1][1])
        else:
            out.append([lo, hi])
    return [(a, b) for a, b in out]


def plan_exchange(ranges: Sequence[Tuple[int, int]],
                  buckets: Optional[Sequence[Tuple[int, int]]] = None) -> List[Tuple[int, int, int]]:
    """One (bucket index, lo, hi) per collective: the trainable span inside each bucket, in the buckets'
    (readiness) order.  A bucket's span runs from its first to its last trainable float -- the frozen gaps
    inside it (zero gradient) ride along rather than splitting the collective; a bucket with nothing
    trainable is skipped.  Without buckets: one span over everything trainable."""
    if not ranges:
        return []
    if not buckets:
        buckets = [(ranges[0][0], ranges[-1][1])]
    plan = []
    for bi, (blo, bhi) in enumerate(buckets):
        inside = [(max(lo, blo), min(hi, bhi)) for lo, hi in ranges if lo < bhi and hi > blo]
        if inside:
            plan.append((bi, inside[0][0], inside[-1][1]))
    return plan


def exchange_mode() -> str:
    """``BTSBOT_AMD_EXCHANGE``: "allreduce" (default: one all-reduce per span) or "rs_ag" (the direct form SURVEY.md
    section 5.8 recommends for xGMI's point-to-point links: reduce-scatter, then all-gather, in place)."""
    import os
    m = os.environ.get("BTSBOT_AMD_EXCHANGE", "allreduce")
    if m not in ("allreduce", "rs_ag"):
        raise ValueError(f"BTSBOT_AMD_EXCHANGE={m!r}: expected 'allreduce' or 'rs_ag'")
    return m


def _span_collectives(flat: torch.Tensor, mode: str, group, async_op: bool):
    """The collectives of ONE span (a 1-D view of the arena), summed over the ranks in place; returns the work handles.
    rs_ag: every rank receives the sum of its 1 / N slice (reduce-scatter), then all slices travel to everyone
    (all-gather); what is left after N equal slices (< N floats) is all-reduced.  On backends without the tensor forms
    (gloo) the same data flow is spelled with reduce-to-owner + broadcast-from-owner per slice."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = flat.numel()
    backend = dist.get_backend(group)
    if mode == "allreduce" or world == 1 or n < world or (backend != "nccl" and flat.device.type != "cpu"):
        # (gloo has no reduce() for device tensors: the one-GPU rehearsals stay on all_reduce)
        return [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)]
    slice_n = n // world
    body = flat[:slice_n * world]
    mine = body[rank * slice_n:(rank + 1) * slice_n]
    works = []
    if backend == "nccl":
        works.append(dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
        works.append(dist.all_gather_into_tensor(body, mine, group=group, async_op=async_op))
    else:
        for r in range(world):
            works.append(dist.reduce(body[r * slice_n:(r + 1) * slice_n], dst=dist.get_global_rank(group, r) if group else r,
                                     op=dist.ReduceOp.SUM, group=group, async_op=async_op))
        for r in range(world):
            works.append(dist.broadcast(body[r * slice_n:(r + 1) * slice_n], src=dist.get_global_rank(group, r) if group else r,
                                        group=group, async_op=async_op))
    if slice_n * world < n:
        works.append(dist.all_reduce(flat[slice_n * world:], op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    return works


class GradExchange:
    """The exchange step of one training iteration over a flat gradient arena.

    ``exchange(grads, wait_bucket)`` issues the collectives of one planned span after the other (``exchange_mode()``:
    one all-reduce per span, or reduce-scatter + all-gather).  On a HIP device the collectives
    go to a side stream: ``wait_bucket(bucket, stream)`` (``btsbot_wait_grad_bucket``) makes that stream wait
    for the bucket's gradients, and the caller's stream waits for every collective before it returns to the
    optimiser.  On CPU tensors (gloo) the calls are synchronous and ``wait_bucket`` is not used.
    """

    def __init__(self, ranges: Sequence[Tuple[int, int]],
                 buckets: Optional[Sequence[Tuple[int, int]]] = None, group=None, mode: Optional[str] = None):
        self.group = group
        self.plan = plan_exchange(ranges, buckets)
        self.mode = mode or exchange_mode()
        self._side = None

    def exchange(self, grads: torch.Tensor,
                 wait_bucket: Optional[Callable[[int, int], None]] = None) -> torch.Tensor:
        if _world(self.group) <= 1 or not self.plan:
            return grads
        import torch.distributed as dist
        if grads.device.type != "cuda":
            for _b, lo, hi in self.plan:
                _span_collectives(grads[lo:hi], self.mode, self.group, async_op=False)
            return grads
        dev = grads.device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        works = []
        with torch.cuda.stream(self._side):
            for b, lo, hi in self.plan:
                if wait_bucket is not None:
                    wait_bucket(b, self._side.cuda_stream)     # the side stream waits for the bucket's kernels
                else:
                    self._side.wait_stream(main)
                works.extend(_span_collectives(grads[lo:hi], self.mode, self.group, async_op=True))
        for w in works:
            w.wait()                                            # the caller's stream waits for the collective
        main.wait_stream(self._side)
        return grads


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """Sum `flat` over ranks in place.  The local gradients are already scaled by 1/B_global
    (btsbot_bce_fwd_bwd's n_global), so the SUM is the gradient of the global-batch mean loss."""
    import torch.distributed as dist
    if _world(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Make every replica start from rank `src`'s parameters (what DataParallel's per-step
    broadcast guaranteed implicitly)."""
    import torch.distributed as dist
    if _world(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat
