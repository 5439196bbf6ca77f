"""N > 1 path on CPU (gloo, world_size 2): batch sharding + the one gradient exchange per step.

The HIP kernels cannot run here; what is checked is the host logic that makes the multi-GPU step
correct by construction (btsbot_amd/parallel.py, Trainer.step): every rank differentiates the SUM of
its shard's losses divided by the GLOBAL batch (btsbot_bce_fwd_bwd's n_global), so that one
all-reduce(SUM) of the flat gradient arena yields exactly the gradient of the global-batch mean loss
that a single process would compute."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from btsbot_amd import parallel


def test_shard_bounds_cover_the_batch():
    for n in (0, 1, 7, 8, 39, 1024):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    from helpers import CONFIGS, seeded_state
    from btsbot_amd.synthetic import synthetic_batch
    from oracle import convnext_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    kind, cfg = CONFIGS["um_nn"]
    cfg = dict(cfg, meta_dropout=0.0)
    n_global = 22                                    # ragged: 11 + 11; try 3-way raggedness below
    _, meta, labels = synthetic_batch(n_global, seed=5)
    # every replica starts from rank 0's parameters
    sd = seeded_state(kind, cfg, seed=3 + rank)
    allk = [k for k, v in sd.items() if v.is_floating_point()]     # parameters AND BN buffers
    keys = [k for k in allk if "running" not in k]
    flat = torch.cat([sd[k].reshape(-1) for k in allk])
    parallel.broadcast_(flat, src=0)
    off = 0
    for k in allk:
        sd[k] = flat[off:off + sd[k].numel()].view_as(sd[k]).clone()
        off += sd[k].numel()
    for k in keys:
        sd[k].requires_grad_(True)
    lo, hi = parallel.shard_bounds(n_global, rank, world)
    # eval-mode BN here so that the shard and the full batch see the same function of the inputs
    # (training-mode BN statistics are per-rank by design, like DataParallel: train.py:238-240)
    logits = O.forward(kind, sd, cfg, None, meta[lo:hi], training=False)
    z, y = logits.reshape(-1), labels[lo:hi].float()
    local = -(2.0 * y * torch.nn.functional.logsigmoid(z) +
              (1 - y) * torch.nn.functional.logsigmoid(-z)).sum() / n_global
    local.backward()
    g = torch.cat([sd[k].grad.reshape(-1) for k in keys])
    parallel.allreduce_mean_(g)
    loss = local.detach().clone()
    dist.all_reduce(loss)
    if rank == 0:
        q.put((g.numpy(), loss.numpy(), flat.numpy()))   # by value: see _exchange_worker
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_single_process():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    from helpers import CONFIGS, seeded_state
    from btsbot_amd.synthetic import synthetic_batch
    from oracle import convnext_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g2, loss2, flat0 = (torch.from_numpy(t) for t in q.get(timeout=240))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the whole batch with rank 0's parameters
    kind, cfg = CONFIGS["um_nn"]
    cfg = dict(cfg, meta_dropout=0.0)
    _, meta, labels = synthetic_batch(22, seed=5)
    sd = seeded_state(kind, cfg, seed=3)
    allk = [k for k, v in sd.items() if v.is_floating_point()]
    keys = [k for k in allk if "running" not in k]
    assert torch.equal(flat0, torch.cat([sd[k].reshape(-1) for k in allk]))   # broadcast worked
    for k in keys:
        sd[k].requires_grad_(True)
    logits = O.forward(kind, sd, cfg, None, meta, training=False)
    ref = O.bce_with_logits(logits, labels.float().unsqueeze(1), 2.0)
    ref.backward()
    g1 = torch.cat([sd[k].grad.reshape(-1) for k in keys])
    assert abs(loss2.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert (g2 - g1).abs().max().item() <= 1e-5 * g1.abs().max().item()


# ---- the exchange step Trainer.step really runs (parallel.trainable_ranges / plan_exchange / GradExchange), on the
#      model's own arena layout and gradient buckets, with frozen tensors so that gaps exist -------------------------
def _freeze_some(model):
    """Freeze stage 1 of the image branch and the metadata BatchNorm affine: trainable ranges with gaps inside a
    bucket (stage 1 sits in the stem + stages 0-1 bucket) and next to the BN buffers."""
    for k, p in model.named_parameters():
        if ".stages.1." in k or k.startswith("metadata_branch.0."):
            p.requires_grad_(False)


def _exchange_worker(rank, world, port, q):
    import sys
    import warnings
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import btsbot_amd
    from btsbot_amd.train import Trainer
    from helpers import CONFIGS
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    kind, cfg = CONFIGS["mm_pico"]
    torch.manual_seed(100 + rank)                       # replicas start DIFFERENT ...
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = getattr(btsbot_amd, kind)(cfg)
    _freeze_some(m)
    before = m._arena.clone()
    tr = Trainer(m, lr=1e-4)                            # ... and the Trainer makes them rank 0's
    after = m._arena.clone()
    ranges, plan = tr.ranges, tr.exchange.plan
    # a fake local gradient arena: rank-dependent everywhere, so untouched entries stay recognisable
    g = torch.arange(m._arena.numel(), dtype=torch.float32) * 1e-6 + (rank + 1)
    tr.exchange.exchange(g)
    out = dict(rank=rank, ranges=ranges, plan=plan, buckets=m._grad_buckets(), g=g,
               arena_changed=bool((before != after).any()), arena=after)
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: v for k, v in out.items() if k != "g" and k != "arena"})
    arenas = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(arenas, after)
    if rank == 0:
        # by value (numpy): a torch tensor in the queue is fetched from THIS process, which may be gone by then
        out = dict(out, g=g.numpy(), arena=None)
        q.put((out, gathered, all(torch.equal(a, arenas[0]) for a in arenas)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
@pytest.mark.parametrize("world", [2, 3])
def test_trainer_exchange_plan_and_broadcast(world, mode, monkeypatch):
    """(mode: one all-reduce per span, or BTSBOT_AMD_EXCHANGE=rs_ag -- reduce-scatter to the slice owners + all-gather,
    spelled with reduce / broadcast on gloo; spans that do not divide by the world size leave a remainder all-reduce)"""
    monkeypatch.setenv("BTSBOT_AMD_EXCHANGE", mode)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, gathered, same_arena = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same_arena                                   # broadcast at Trainer construction
    ranges, plan, buckets, g = out["ranges"], out["plan"], out["buckets"], torch.from_numpy(out["g"])
    assert all(o["ranges"] == ranges and o["plan"] == plan for o in gathered)    # same plan on every rank
    n = g.numel()
    assert len(buckets) == 3 and buckets[0][1] == n and buckets[2][0] == 0      # readiness order: tail first
    assert len(ranges) >= 3                                                     # the frozen tensors left gaps
    # every trainable float lies in exactly one planned span, spans follow the bucket order and stay inside them
    assert [b for b, _lo, _hi in plan] == sorted(b for b, _lo, _hi in plan)
    for b, lo, hi in plan:
        assert buckets[b][0] <= lo < hi <= buckets[b][1]
    trainable = torch.zeros(n, dtype=torch.bool)
    for lo, hi in ranges:
        trainable[lo:hi] = True
    covered = torch.zeros(n, dtype=torch.int32)
    for _b, lo, hi in plan:
        covered[lo:hi] += 1
    assert bool((covered[trainable] == 1).all()) and int(covered.max()) == 1
    # reduced entries hold the SUM over ranks, everything outside the spans is still rank 0's local value
    base = torch.arange(n, dtype=torch.float32) * 1e-6
    total = base * world + sum(r + 1 for r in range(world))
    red = covered == 1
    assert torch.allclose(g[red], total[red], rtol=1e-6)
    assert torch.equal(g[~red], (base + 1)[~red])
    assert bool((~red).any())                           # e.g. the BatchNorm buffers between trainable tensors
