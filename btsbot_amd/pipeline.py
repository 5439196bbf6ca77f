"""Streaming scorer: keeps `depth` forward passes in flight on alternating HIP streams.

One forward pass ends in kernels that leave most of the chip idle (the 1x1 stage and the classifier head: a few
hundred workgroups of latency-bound work), and a stream runs its kernels strictly one after the other, each
boundary costing ~3 us.  Scoring a night's alerts is a sequence of independent batches, so consecutive batches
are issued on different streams -- each stream with its own replica of the model (own workspace, own packed
weights; the parameters are a few tens of MB against 288 GB of HBM) -- and the tail of batch n runs under the
front of batch n + 1.  Measured on MI355X, mm_ConvNeXt-pico bf16, 1024-alert batches: 0.330 ms per batch as plain calls,
0.313 with two batches in flight, 0.301 with three (the default; four: 0.302).

This is the reference's scoring loop (`for triplets, metadata in loader: model(triplets, metadata)`,
/root/reference/btsbot/val.py:128-157 and inference_example.py:75-91) with the per-batch calls overlapped; a single
`model(x)` call keeps PyTorch's stream semantics and is not affected.

    scorer = ScoreStream(model)                 # depth=3: three batches in flight
    for logits in scorer.map(batches):          # batches: iterable of (triplets, metadata) tuples
        ...

Synchronisation is kept off the GPU's queues: a result is handed out once its batch has FINISHED (the host waits for
the batch's stream; with `depth` batches queued the GPU never runs dry), so it is valid on any stream without a
stream-side wait -- cross-stream waits cost ~10 us of barrier packets each on this stack, four of them per batch
ate the whole gain (0.32 -> 0.39 ms per batch), and a host-side wait right behind the batch just queued (lag =
depth) cost half of it; map() therefore keeps the host `lag` batches (default 16) ahead.  For the same reason the side stream waits for the caller's stream
only when the inputs may still be in flight there (`inputs_ready=False`, the default); a loader that hands over
finished tensors (synchronous copies, pinned-memory prefetch with its own sync) passes `inputs_ready=True`.
"""
from __future__ import annotations

import collections
from typing import Iterable, Iterator, Tuple

import torch

from . import _lib

__all__ = ["ScoreStream"]


def _replica(model):
    cfg = getattr(model, "_init_config", None)
    if cfg is None:
        raise NotImplementedError(f"ScoreStream(depth > 1) cannot replicate a {type(model).__name__}")
    # (frozen_fusion reads its branches' checkpoints when constructed: the replica takes them from the state dict)
    twin = type(model)(dict(cfg, pretrained=False, skip_load_state=True), precision=model.precision)
    twin.load_state_dict(model.state_dict())
    dev = next(model.parameters()).device
    return twin.to(dev).eval()


_S2P_HINT = int(__import__("os").environ.get("BTSBOT_AMD_S2P_HINT", "7"))   # (developer A/B: 4, 5 or 7; 0 = by rounds)
if _S2P_HINT not in (0, 4, 5, 7):   # (the library would reject the value and the hint would silently do nothing)
    raise ValueError(f"BTSBOT_AMD_S2P_HINT={_S2P_HINT}: stage2p keeps 4, 5 or 7 alerts per workgroup (0: its own choice)")


class ScoreStream:
    def __init__(self, model, depth: int = 3, inputs_ready: bool = False):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if model.training:
            raise RuntimeError("ScoreStream scores with an eval-mode model; call model.eval() first")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("btsbot_amd: ScoreStream needs the model on an AMD GPU ('cuda'); there is no CPU fallback")
        self.models = [model] + [_replica(model) for _ in range(depth - 1)]
        self.device = dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.inputs_ready = inputs_ready
        self._next = 0
        self._order_after_caller()

    def _order_after_caller(self):
        # the replicas' parameters were written on the caller's stream (load_state_dict, .to(device)); their first
        # btsbot_pack_params runs on a side stream: order the two whatever `inputs_ready` says
        cur = torch.cuda.current_stream(self.device)
        for side in self.streams:
            side.wait_stream(cur)

    def refresh(self):
        """Copy the first model's parameters into the replicas (after they changed).  The caller's stream first waits for
        every side stream: a replica's queued work (a forward, or the arena -> mirror copy of its pending re-pack) must
        not overlap the overwrite of its parameters."""
        cur = torch.cuda.current_stream(self.device)
        for side in self.streams:
            cur.wait_stream(side)
        sd = self.models[0].state_dict()
        for twin in self.models[1:]:
            twin.load_state_dict(sd)
        self._order_after_caller()

    def submit(self, *inputs: torch.Tensor):
        """Enqueue one batch; returns a ticket for result().  The ticket keeps the input tensors alive until result()
        has seen the batch finish (they were allocated on the caller's stream and are read on a side stream: dropped
        earlier, the caching allocator could hand their memory to the caller's next batch while this one is still
        queued); the caller must not MODIFY them until then."""
        k = self._next
        self._next = (k + 1) % len(self.models)
        side = self.streams[k]
        if not self.inputs_ready:
            side.wait_stream(torch.cuda.current_stream(self.device))  # the inputs are ready in the caller's order
        m = self.models[k]
        hinted = len(self.models) > 1 and getattr(m, "_handle", None) is not None
        if hinted:
            # several forwards in flight: the stage-2 kernel keeps 7 alerts per workgroup at every batch size and so
            # leaves ~40 % of the CUs to the other stream's kernels (+6 % through this loop at 1024 alerts; a lone
            # model(...) call is 4 % slower that way, so the hint is taken back right after the launches are queued)
            _lib.check(_lib.lib().btsbot_set_option(m._handle.ptr, b"stage2p_alerts", _S2P_HINT), "btsbot_set_option")
        with torch.cuda.stream(side), torch.no_grad():
            try:
                out = m(*inputs)
            finally:
                if hinted:
                    _lib.lib().btsbot_set_option(m._handle.ptr, b"stage2p_alerts", 0)
            done = torch.cuda.Event()
            done.record(side)
        return out, done, inputs

    def result(self, ticket) -> torch.Tensor:
        """The batch's logits; returns once the batch has finished on the GPU (host-side wait), so the tensor is valid
        on every stream."""
        out, done, _inputs = ticket
        done.synchronize()
        # `out` was allocated while a side stream was current: tell the allocator that the caller's stream uses it too,
        # or the block could be handed to a later forward on that side stream while a caller-stream kernel still reads it
        out.record_stream(torch.cuda.current_stream(self.device))
        return out

    def map(self, batches: Iterable[Tuple[torch.Tensor, ...]], lag: int = 16) -> Iterator[torch.Tensor]:
        """Score an iterable of input tuples, results in order.  The host stays `lag` batches ahead of the oldest
        unfinished one: every batch is followed by a completion event, and a result is handed out (after a host-side
        wait on its event) once `lag` newer batches are queued.  A short lag makes the loop sensitive to host jitter (with 4
        batches = 1.3 ms of queued work one descheduling of the host thread empties the queues); the price of a long one
        is `lag` output tensors and input tuples kept alive."""
        lag = max(lag, len(self.models))
        pending = collections.deque()
        for inputs in batches:
            if not isinstance(inputs, (tuple, list)):
                inputs = (inputs,)
            pending.append(self.submit(*inputs))
            if len(pending) > lag:
                yield self.result(pending.popleft())
        while pending:
            yield self.result(pending.popleft())
