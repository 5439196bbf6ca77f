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
        q.put((g, loss, flat))
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
    g2, loss2, flat0 = q.get(timeout=240)
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
