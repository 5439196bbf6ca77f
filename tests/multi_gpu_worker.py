"""One rank of the N > 1 GPU test (tests/test_gpu_multi.py starts N of these as child processes; it is not a test
module itself).  Environment: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, BTSBOT_TEST_BACKEND (nccl = RCCL, one GPU
per rank; gloo = every rank on cuda:0, the rehearsal a one-GPU box can run), BTSBOT_TEST_OUT (rank 0's result file).

What it checks is the exchange step that replaces torch.nn.parallel.DataParallel
(/root/reference/btsbot/train.py:238-240), on the GPU path:
  1. sharded + exchanged gradients == a single-process pass over the whole batch (image-only ConvNeXt-pico, dropout 0,
     64 alerts per rank, fp32 mode: no BatchNorm1d, whose batch statistics are per rank by design);
  2. the gradient buckets' events fire before their all-reduce: the exchanged gradient equals the reference on EVERY
     bucket, also on a second step whose arena starts out holding the first step's exchanged sums (a collective that ran
     ahead of its bucket would reduce those);
  3. two Trainer.step calls on mm_ConvNeXt (BatchNorm1d + dropout, per-rank masks) leave all replicas bit-identical.
"""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("BTSBOT_TEST_WATCHDOG", "240")), exit=True)   # a hung collective
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("BTSBOT_TEST_BACKEND", "nccl")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import btsbot_amd
    from btsbot_amd import parallel
    from btsbot_amd.train import Trainer
    from btsbot_amd.synthetic import synthetic_batch
    from helpers import CONFIGS, seeded_state
    res = {"backend": backend, "world": world}
    # a one-rank group per rank (every rank takes part in every new_group call): the single-process reference pass on
    # rank 0 must not enter the world's collectives
    solo = [dist.new_group([r]) for r in range(world)][rank]

    # ---- 1 + 2: gradients of a sharded step against the whole batch in one process ---------------------------
    per_rank = 64
    kind, cfg = CONFIGS["convnext"]
    cfg = dict(cfg, dropout=0.0)
    img, _meta, lab = synthetic_batch(per_rank * world, seed=11)
    img, lab = img.to(dev), lab.to(dev)

    def build(seed):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = getattr(btsbot_amd, kind)(cfg, precision="f32")
        m.load_state_dict(seeded_state(kind, cfg, seed=seed))
        return m.to(dev).train()

    m = build(3 + rank)                                  # replicas start different: the Trainer makes them rank 0's
    tr = Trainer(m, lr=1e-4)
    buckets = m._grad_buckets()
    lo, hi = parallel.shard_bounds(per_rank * world, rank, world)
    loss, g = tr.gradients(img[lo:hi].contiguous(), None, lab[lo:hi].contiguous())
    g = g.clone()
    # the same step again: the gradient arena now holds the previous step's EXCHANGED sums, so a collective that did
    # not wait for its bucket's event would reduce stale values (world x too large) instead of this step's gradients
    _l2, g2 = tr.gradients(img[lo:hi].contiguous(), None, lab[lo:hi].contiguous())
    g2 = g2.clone()
    tot = loss.detach().clone().reshape(1)
    dist.all_reduce(tot)
    torch.cuda.synchronize(dev)
    if rank == 0:
        ref_m = build(3)
        ref_tr = Trainer(ref_m, lr=1e-4, group=solo)
        rloss, rg = ref_tr.gradients(img, None, lab, global_batch=per_rank * world, exchange=False)
        torch.cuda.synchronize(dev)
        scale = rg.abs().max().item()
        res["grad_max_abs"] = scale
        res["grad_err"] = (g - rg).abs().max().item() / scale
        res["grad_err_repeat"] = (g2 - rg).abs().max().item() / scale
        res["loss_err"] = abs(tot.item() - rloss.item())
        res["bucket_err"] = [(g[a:b] - rg[a:b]).abs().max().item() / scale for a, b in buckets if b > a]
        res["n_buckets"] = len(buckets)
        res["plan"] = [list(p) for p in tr.exchange.plan]
    # ---- 2b (RCCL only): the same exchange through the C ABI (btsbot_allreduce_grads) on a raw communicator --------
    if backend == "nccl":
        from btsbot_amd.rccl import RcclComm
        box = [RcclComm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = RcclComm(rank, world, box[0])
        tr_c = Trainer(m, lr=1e-4, rccl_comm=comm)
        _l3, g3 = tr_c.gradients(img[lo:hi].contiguous(), None, lab[lo:hi].contiguous())
        g3 = g3.clone()
        torch.cuda.synchronize(dev)
        if rank == 0:
            res["grad_err_c_abi"] = (g3 - rg).abs().max().item() / scale
        # ... and in its reduce-scatter + all-gather form (btsbot_set_option(h, "exchange", 1)), then the same form
        # through torch.distributed (BTSBOT_AMD_EXCHANGE=rs_ag)
        from btsbot_amd import _lib
        _lib.check(_lib.lib().btsbot_set_option(m._handle.ptr, b"exchange", 1), "set_option")
        _l4, g4 = tr_c.gradients(img[lo:hi].contiguous(), None, lab[lo:hi].contiguous())
        g4 = g4.clone()
        _lib.check(_lib.lib().btsbot_set_option(m._handle.ptr, b"exchange", 0), "set_option")
        tr_rs = Trainer(m, lr=1e-4)
        tr_rs.exchange.mode = "rs_ag"
        _l5, g5 = tr_rs.gradients(img[lo:hi].contiguous(), None, lab[lo:hi].contiguous())
        g5 = g5.clone()
        torch.cuda.synchronize(dev)
        if rank == 0:
            res["grad_err_c_abi_rs_ag"] = (g4 - rg).abs().max().item() / scale
            res["grad_err_rs_ag"] = (g5 - rg).abs().max().item() / scale
        comm.destroy()
    del m, tr

    # ---- 3: full training steps with BatchNorm1d and dropout: replicas stay identical ------------------------
    kind, cfg = CONFIGS["mm_pico"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mm = getattr(btsbot_amd, kind)(cfg, precision="bf16")
    mm.load_state_dict(seeded_state(kind, cfg, seed=5 + rank))
    mm = mm.to(dev).train()
    tr = Trainer(mm, lr=1e-3, betas=(0.99, 0.99), pos_weight=2.0, dropout_seed=1)
    img, meta, lab = synthetic_batch(32 * world, seed=12)
    lo, hi = parallel.shard_bounds(32 * world, rank, world)
    for _ in range(2):
        l = tr.step(img[lo:hi].to(dev), meta[lo:hi].to(dev), lab[lo:hi].to(dev))
    torch.cuda.synchronize(dev)
    arena = mm._arena.detach().clone()
    # trainable entries must agree bit for bit; BatchNorm1d running statistics are per rank (DataParallel semantics)
    mask = torch.zeros_like(arena, dtype=torch.bool)
    for a, b in tr.ranges:
        mask[a:b] = True
    mine = torch.where(mask, arena, torch.zeros_like(arena))
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    same = torch.tensor([1.0 if torch.equal(mine, ref) else 0.0], device=dev)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if rank == 0:
        res["replicas_identical_after_steps"] = bool(same.item() == 1.0)
        res["loss_finite"] = bool(torch.isfinite(l).item())
        with open(os.environ["BTSBOT_TEST_OUT"], "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
