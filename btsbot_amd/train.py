"""Training step of the classifier on MI355X: the hot loop of /root/reference/btsbot/train.py:481-566
(zero_grad -> forward -> BCEWithLogitsLoss(pos_weight) -> backward -> AdamW.step) plus the optimiser
and schedule set-up of train.py:242-260, without autograd, host syncs or DataParallel.

    trainer = Trainer(model, lr=1e-4, betas=(0.99, 0.99), pos_weight=w)   # model.train() already
    for images, meta, labels in batches:                                    # device tensors
        loss = trainer.step(images, meta, labels)        # returns a device scalar, no .item()
    trainer.scheduler_step()                                                # once per epoch

What is trained follows ``requires_grad`` exactly as the reference does (train.py:224-236): for
``frozen_fusion`` only ``combined_head``; otherwise everything, ConvNeXt image branch included.  With torch.distributed initialised, every rank passes its
own shard and the flat gradient arena is all-reduced once per step (parallel.py).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import torch

from . import _lib, parallel


def lr_schedule(lr: float, epochs: int, warmup_epochs: int) -> Sequence[float]:
    """Learning rate of epoch 0..epochs-1 under train.py:249-260's
    SequentialLR[LinearLR(0.01 -> 1 over warmup), CosineAnnealingLR(T_max=max(1, epochs-warmup),
    eta_min=0.01*lr)], stepped once per epoch -- produced by torch's own schedulers so that
    version quirks (warmup 0 leaves the LR at 0.01*lr on torch 2.10) are reproduced, not guessed."""
    import warnings
    prm = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([prm], lr=lr)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sched = torch.optim.lr_scheduler.SequentialLR(
            opt,
            schedulers=[
                torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, total_iters=warmup_epochs),
                torch.optim.lr_scheduler.CosineAnnealingLR(
                    opt, T_max=max(1, epochs - warmup_epochs), eta_min=lr * 0.01),
            ],
            milestones=[warmup_epochs])
        out = []
        for _ in range(epochs):
            out.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
    return out


_EAGER_REPACK = os.environ.get("BTSBOT_AMD_EAGER_REPACK", "1") != "0"


class Trainer:
    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, pos_weight: float = 1.0, epochs: int = 1,
                 warmup_epochs: int = 0, group=None, dropout_seed: int = 0, rccl_comm=None):
        self.model = model
        self.base_lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.pos_weight = float(pos_weight)
        self.group = group
        self.lrs = list(lr_schedule(lr, max(1, epochs), warmup_epochs))
        self.epoch = 0
        self.t = 0                                   # AdamW step count
        comb, meta, image = model._slot_groups()
        self.need_image = any(t.requires_grad for t, *_ in image)
        self.need_meta = any(t.requires_grad for t, *_ in meta)
        # contiguous arena ranges of the trainable tensors (padding between tensors has zero grad)
        self.ranges = parallel.trainable_ranges(
            (off, numel) for t, off, numel, _s in image + meta + comb if t.requires_grad)
        # the exchange step: one all-reduce per gradient bucket, in the order btsbot_backward() finishes them
        self.exchange = parallel.GradExchange(self.ranges, model._grad_buckets(), group)
        self.exp_avg = None
        self.exp_avg_sq = None
        import torch.distributed as dist
        self.rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
        self.dropout_seed = int(dropout_seed)
        self._rng = None
        # the exchange step through the C ABI (btsbot_allreduce_grads) on a raw RCCL communicator (btsbot_amd.rccl.RcclComm)
        # instead of torch.distributed: what a host that binds only libbtsbot_hip.so does.  The replicas must then be
        # made equal by the caller (no process group to broadcast over)
        self.rccl = rccl_comm
        if rccl_comm is not None:
            self.rank = rccl_comm.rank
        # replicas start from rank 0's parameters and buffers (what DataParallel's per-step broadcast did
        # implicitly at train.py:238-240); from here on identical gradients keep them identical
        if parallel._world(group) > 1:
            parallel.broadcast_(model._arena, 0, group)
            model.mark_weights_dirty()

    @property
    def lr(self) -> float:
        return self.lrs[min(self.epoch, len(self.lrs) - 1)]

    def scheduler_step(self):
        """train.py:332 -- once per epoch."""
        self.epoch += 1

    def gradients(self, images: Optional[torch.Tensor], meta: Optional[torch.Tensor], labels: torch.Tensor,
                  global_batch: Optional[int] = None, exchange: bool = True):
        """Forward + BCE + backward (+ the exchange step when `exchange` and more than one rank) on this rank's shard,
        without the optimiser update: returns (sum_i loss_i / global_batch as a device scalar, the flat gradient arena
        -- a buffer the library reuses on the next call).  `step` is this followed by AdamW; the multi-GPU tests call it
        to compare sharded + exchanged gradients with a single-process pass over the whole batch."""
        m = self.model
        if not m.training:
            raise RuntimeError("Trainer.step: put the model in train mode first (model.train())")
        images, meta, batch, dev = m._check_inputs(images, meta)
        if labels.numel() != batch:
            raise ValueError(f"Trainer.step: {labels.numel()} labels for a batch of {batch} alerts")
        if batch < 2 and m._cfg_args["n_meta"] > 0:
            # nn.BatchNorm1d in train mode: "Expected more than 1 value per channel when training"
            raise ValueError("Trainer.step: BatchNorm1d batch statistics need more than one alert per rank")
        world = self.rccl.world if self.rccl is not None else parallel._world(self.group)
        n_global = int(global_batch) if global_batch is not None else batch * world
        if self._rng is None or self._rng.device != dev:
            # every rank draws its own dropout masks (seed + rank)
            self._rng = torch.Generator(device=dev)
            self._rng.manual_seed(self.dropout_seed + 7919 * self.rank)
        with torch.no_grad():
            masks = m._dropout_masks(batch, dev, self._rng if world > 1 else None)
            logits = m._forward_train_raw(images, meta, masks, self.need_image).reshape(-1)
            y = labels.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
            loss = torch.zeros(1, dtype=torch.float32, device=dev)
            dl = torch.empty_like(logits)
            L = _lib.lib()
            with torch.cuda.device(dev):
                st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                _lib.check(L.btsbot_bce_fwd_bwd(C.c_void_p(logits.data_ptr()), C.c_void_p(y.data_ptr()),
                                                self.pos_weight, batch, n_global,
                                                C.c_void_p(loss.data_ptr()), C.c_void_p(dl.data_ptr()),
                                                st), "btsbot_bce_fwd_bwd")
            grads = m._backward_raw(dl, self.need_meta, self.need_image)
            if self.rccl is not None and exchange:
                plan = self.exchange.plan
                n = len(plan)
                bk = (C.c_int32 * n)(*[b for b, _lo, _hi in plan])
                lo = (C.c_int64 * n)(*[a for _b, a, _hi in plan])
                hi = (C.c_int64 * n)(*[z for _b, _lo, z in plan])
                with torch.cuda.device(dev):
                    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                    _lib.check(L.btsbot_allreduce_grads(m._handle.ptr, C.c_void_p(self.rccl.ptr),
                                                        C.c_void_p(grads.data_ptr()), n, bk, lo, hi, st),
                               "btsbot_allreduce_grads")
            elif world > 1 and exchange:
                # the one exchange of the step (local gradients are already scaled by 1 / n_global)
                self.exchange.exchange(grads, m._wait_grad_bucket)
            self.last_logits = logits
        return loss[0] / n_global, grads

    def step(self, images: Optional[torch.Tensor], meta: Optional[torch.Tensor],
             labels: torch.Tensor, global_batch: Optional[int] = None, exchange: bool = True) -> torch.Tensor:
        """One optimisation step on this rank's shard; returns sum_i loss_i / global_batch as a
        device scalar (the rank's contribution to the global mean loss).  (`exchange=False` leaves the all-reduce
        out: bench.py times the step both ways to report what the collective costs in wall time.)"""
        m = self.model
        loss, grads = self.gradients(images, meta, labels, global_batch, exchange)
        dev = grads.device
        with torch.no_grad():
            if self.exp_avg is None:
                self.exp_avg = torch.zeros_like(m._arena)
                self.exp_avg_sq = torch.zeros_like(m._arena)
            self.t += 1
            L = _lib.lib()
            with torch.cuda.device(dev):
                st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for lo, hi in self.ranges:
                    _lib.check(L.btsbot_adamw_step(
                        C.c_void_p(m._arena[lo:hi].data_ptr()), C.c_void_p(grads[lo:hi].data_ptr()),
                        C.c_void_p(self.exp_avg[lo:hi].data_ptr()),
                        C.c_void_p(self.exp_avg_sq[lo:hi].data_ptr()), hi - lo, self.lr,
                        self.betas[0], self.betas[1], self.eps, self.wd, self.t, st),
                        "btsbot_adamw_step")
            m.mark_weights_dirty()
            if _EAGER_REPACK and self.need_image:
                # the operand images of the NEXT step, queued now: the library packs them on its side stream, under the
                # next step's mask draws and stem instead of in front of its first consumer (BTSBOT_AMD_EAGER_REPACK=0: A/B).
                # The next training forward trusts this pack while the parameters' tensor versions are unchanged: code
                # that writes the arena BEHIND torch's back between two steps (a raw C-ABI kernel, `.data` edits, a
                # custom exchange) must call model.mark_weights_dirty() afterwards, or the step runs on stale images.
                if getattr(m, "_reserved_image", False):
                    with torch.cuda.device(dev):
                        m._prepare(dev, int(self.last_logits.numel()), train_only=True)   # (one logit per alert)
        return loss


def train_epoch(trainer: Trainer, dataset, epoch_metrics: bool = True):
    """train.py:481-566 for a DeviceDataset: one pass over the shuffled, augmented batches with
    ``Trainer.step`` and -- like the reference, which recomputes them on the concatenated logits
    (train.py:550-558) -- the epoch's BCE loss and accuracy, evaluated once on the device at the end
    instead of a ``.item()`` per batch.  Returns (epoch_loss, epoch_accuracy) as Python floats."""
    from .val import device_metrics
    logits, labels = [], []
    for batch in dataset:
        if len(batch) == 3:
            images, meta, y = batch
        elif dataset.images is not None:
            (images, y), meta = batch, None
        else:
            (meta, y), images = batch, None
        trainer.step(images, meta, y)
        if epoch_metrics:
            logits.append(trainer.last_logits)
            labels.append(y)
    trainer.scheduler_step()                              # train.py:332
    if not epoch_metrics or not logits:
        return float("nan"), float("nan")
    loss, acc = device_metrics(torch.cat(logits), torch.cat(labels), trainer.pos_weight)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(trainer.group) > 1:
        both = torch.stack([loss, acc])                   # equal shards: the mean of the ranks' means
        dist.all_reduce(both, group=trainer.group)
        loss, acc = both / dist.get_world_size(trainer.group)
    return loss.item(), acc.item()


def fit(trainer: Trainer, train_set, val_images, val_metadata, val_labels, model_dir: str,
        epochs: int, patience: int = 10, val_batch_size: int = 1024, config: Optional[dict] = None):
    """The epoch loop of train.py:303-352 on the device: per epoch ``train_epoch`` -> ``latest_model.pth`` ->
    validation pass (val.py's loop, on the live model instead of re-instantiating it from the file) ->
    scheduler step -> ``best_model.pth`` when the validation loss improved by at least 0.5 % ->
    early stopping after ``patience`` epochs without improvement.  Checkpoints are state dicts only, like
    the reference's (no optimiser state); ``report.json`` carries ``config`` for
    to_HF.prep_config, the history and the best epoch's alert-level ``val_summary``.  Returns the run history dict (train/val loss and accuracy per epoch)."""
    import json
    import os
    import numpy as np
    from .to_HF import cpu_state_dict
    from .val import run_val_tensors
    import torch.distributed as dist
    main = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0   # rank 0 writes the files
    if main:
        os.makedirs(model_dir, exist_ok=True)
    hist = {k: np.zeros(epochs) for k in ("train_loss", "train_accuracy", "val_loss", "val_accuracy")}
    best_raw_preds = best_val_labels = None
    since = 0
    done = 0
    for epoch in range(epochs):
        # (train_epoch also steps the LR schedule; the reference steps it after validation, with the same
        #  effect on the next epoch's learning rate)
        tl, ta = train_epoch(trainer, train_set)
        hist["train_loss"][epoch], hist["train_accuracy"][epoch] = tl, ta
        if main:
            torch.save(cpu_state_dict(trainer.model), os.path.join(model_dir, "latest_model.pth"))
        vl, va, raw, lab = run_val_tensors(trainer.model, val_images, val_metadata, val_labels,
                                           batch_size=val_batch_size, pos_weight=trainer.pos_weight)
        trainer.model.train()
        hist["val_loss"][epoch], hist["val_accuracy"][epoch] = vl, va
        done = epoch + 1
        prev_best = min([np.inf] + list(hist["val_loss"][:epoch]))
        if 1.005 * vl < prev_best:                                   # train.py:334-336
            if main:
                torch.save(cpu_state_dict(trainer.model), os.path.join(model_dir, "best_model.pth"))
            best_raw_preds, best_val_labels = np.copy(raw), np.copy(lab)
            since = 0
        else:
            since += 1
            if since >= patience:                                    # train.py:350-352
                break
    out = {k: v[:done].tolist() for k, v in hist.items()}
    # the alert-level numbers of the reference's val_summary (val.py:178-218 -> utils.make_report) for the best epoch;
    # its figure and the per-source policy metrics need the candidate table and stay out of scope
    from .val import alert_summary
    summary = alert_summary(best_raw_preds, best_val_labels) if best_raw_preds is not None else {}
    if main:
        with open(os.path.join(model_dir, "report.json"), "w") as f:
            json.dump({"train_config": dict(config or {}), "Training history": out, "val_summary": summary}, f,
                      indent=2)
    out["best_raw_preds"], out["best_val_labels"] = best_raw_preds, best_val_labels
    out["val_summary"] = summary
    return out


def run_training(config: dict, data_base_dir: str = "", run_name: str = "testing", device="cuda",
                 precision: str = "bf16", models_root: str = "models"):
    """train.py:75-440 (``run_training(config)``) on this framework, minus WandB and the diagnostic figure: seeds,
    the split files (``data.load_split``), the model by name with the frozen_fusion freezing rule
    (train.py:224-236), AdamW(lr, betas=(beta_1, beta_2)) under the warm-up + cosine schedule, BCE with
    pos_weight = N_neg / N_pos of the training split (train.py:211-212), the epoch loop with latest / best
    checkpoints and early stopping (``fit``), and the report.  Under ``torch.distributed`` (one process per GPU,
    launched by torchrun) every rank trains its contiguous shard of each global batch and the gradients meet in one
    all-reduce per step; rank 0 writes the files.  Config keys are the reference's: model_name, epochs,
    batch_size, learning_rate, warmup_epochs, beta_1, beta_2, patience, random_seed, train_data_version, N_max,
    metadata_cols, data_aug_*, plus the model's own.  Returns (history dict, model_dir)."""
    import numpy as np
    import torch.distributed as dist
    from . import architectures
    from .data import DeviceDataset, load_split
    model_name = config["model_name"]
    epochs, batch_size = int(config["epochs"]), int(config["batch_size"])
    lr = float(config["learning_rate"])                    # "sometimes WandB makes LR a string" (train.py:85)
    warmup = int(config.get("warmup_epochs", 0))
    betas = (float(config["beta_1"]), float(config["beta_2"]))
    patience, seed = int(config["patience"]), int(config["random_seed"])
    version, n_str = config["train_data_version"], f"_N{config.get('N_max', 100)}"
    dev = torch.device(device)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    timg, tmeta, tlab, _ = load_split(data_base_dir, config, "train")
    vimg, vmeta, vlab, _ = load_split(data_base_dir, config, "val")
    try:
        model_type = getattr(architectures, model_name)
    except AttributeError:
        raise ValueError(f"Could not find model of name {model_name}") from None
    model = model_type(config, precision=precision).to(dev).train()
    if model_name == "frozen_fusion":                      # only the combined head is trained
        for p in list(model.image_branch.parameters()) + list(model.meta_branch.parameters()):
            p.requires_grad = False
        for p in model.combined_head.parameters():
            p.requires_grad = True
        for m in model._image_bn_modules():                # a MaxViT branch is served in eval mode only
            m.eval()
    else:
        for p in model.parameters():
            p.requires_grad = True
    gen = torch.Generator(device=dev).manual_seed(seed)    # the same permutations on every rank
    train_set = DeviceDataset(timg, tmeta, tlab, batch_size, config=config, device=dev, generator=gen,
                              shard=(rank, world) if world > 1 else None)
    trainer = Trainer(model, lr=lr, betas=betas, pos_weight=train_set.pos_weight, epochs=epochs,
                      warmup_epochs=warmup)
    model_dir = os.path.join(models_root, f"{model_name}_{version}{n_str}_{dev.type}", run_name) + "/"
    hist = fit(trainer, train_set, vimg, vmeta, vlab, model_dir, epochs=epochs, patience=patience,
               val_batch_size=batch_size, config=config)
    return hist, model_dir
