"""Validation pass and scalar metrics on the device (/root/reference/btsbot/val.py:31-170).

``run_val_tensors`` is the evaluate loop of val.py:117-168 for a split that is already in memory:
forward in batches (no shuffling, no augmentation), then BCEWithLogitsLoss(pos_weight) over ALL logits
and the accuracy of ``sigmoid(logits) > 0.5`` -- accumulated by ``btsbot_eval_metrics`` without a
per-batch ``.item()``.  Returns what the reference returns: (loss, accuracy, raw_preds, labels), the
last two as numpy arrays.  The model object is reused (the reference re-instantiates it and reloads
``best_model.pth`` on every call, val.py:64-74).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .data import DeviceDataset


def device_metrics(logits: torch.Tensor, labels: torch.Tensor, pos_weight: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(mean BCE-with-logits loss, accuracy) as device scalars; logits [N] or [N,1], labels 0/1."""
    z = logits.reshape(-1).to(torch.float32).contiguous()
    y = labels.reshape(-1).to(device=z.device, dtype=torch.float32).contiguous()
    if z.device.type != "cuda":
        raise RuntimeError("btsbot_amd.val.device_metrics runs on the GPU; there is no CPU fallback")
    if z.numel() != y.numel():
        raise ValueError("logits / labels length mismatch")
    out = torch.zeros(2, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        st = torch.cuda.current_stream(z.device).cuda_stream
        _lib.check(_lib.lib().btsbot_eval_metrics(C.c_void_p(z.data_ptr()), C.c_void_p(y.data_ptr()),
                                                  float(pos_weight), z.numel(),
                                                  C.c_void_p(out.data_ptr()), C.c_void_p(st)),
                   "btsbot_eval_metrics")
    n = max(z.numel(), 1)
    return out[0] / n, out[1] / n


def run_val_tensors(model, images, metadata, labels, batch_size: int = 1024,
                    pos_weight: Optional[float] = None, device="cuda"):
    """val.py:117-168.  ``pos_weight`` defaults to N_neg / N_pos of this split (val.py:60-62)."""
    ds = DeviceDataset(images, metadata, labels, batch_size, device=device, shuffle=False,
                       drop_last=False, augment=False, check_nan=False)
    pw = ds.pos_weight if pos_weight is None else float(pos_weight)
    was_training = model.training
    model.eval()
    logits = []
    with torch.no_grad():
        for batch in ds:
            if ds.images is not None and ds.metadata is not None:
                logits.append(model(image_input=batch[0], metadata_input=batch[1]))
            else:
                logits.append(model(input_data=batch[0]))
    model.train(was_training)
    all_logits = torch.cat(logits, dim=0) if logits else torch.empty(0, 1, device=ds.device)
    loss, acc = device_metrics(all_logits, ds.labels, pw)
    raw_preds = torch.sigmoid(all_logits).squeeze(1).cpu().numpy()
    return loss.item(), acc.item(), raw_preds, ds.labels.float().cpu().numpy()


def alert_summary(raw_preds, labels) -> dict:
    """The alert-level part of the reference's validation summary (val.py:178-218; what train.py:404-411 logs and
    report.json keeps under ``val_summary``): confusion counts at the 0.5 threshold (``np.rint``: 0.5 rounds to 0),
    ``bts_acc`` / ``notbts_acc`` / ``bal_acc``, ``alert_precision`` / ``alert_recall`` (-999.0 when there is no true
    positive or no true negative, as there) and ``roc_auc`` (area under sklearn's ``roc_curve``: the rank statistic
    with tied scores sharing their average rank).  Everything is reduced on the tensors' device; one host read."""
    p = torch.as_tensor(raw_preds).reshape(-1).to(torch.float64)
    y = torch.as_tensor(labels, device=p.device).reshape(-1).to(torch.float64)
    if p.numel() != y.numel():
        raise ValueError("raw_preds / labels length mismatch")
    pred = torch.round(p)                                  # half-to-even, as np.rint
    tp = ((pred == 1) & (y == 1)).sum()
    tn = ((pred == 0) & (y == 0)).sum()
    fp = ((pred == 1) & (y == 0)).sum()
    fn = ((pred == 0) & (y == 1)).sum()
    # ROC AUC = P(score_pos > score_neg) + 0.5 P(tie): average ranks of the sorted scores
    order = torch.argsort(p, stable=True)
    ps = p[order]
    uniq, inv, cnt = torch.unique_consecutive(ps, return_inverse=True, return_counts=True)
    last = torch.cumsum(cnt, 0).to(torch.float64)          # rank of the last member of each tie group (1-based)
    avg_rank = (last - (cnt.to(torch.float64) - 1) / 2)[inv]
    n_pos, n_neg = y.sum(), (1 - y).sum()
    rank_sum = (avg_rank * y[order]).sum()
    vals = torch.stack([tp, tn, fp, fn, n_pos, n_neg, rank_sum]).to(torch.float64).cpu().tolist()
    tp, tn, fp, fn, n_pos, n_neg, rank_sum = vals
    nan = float("nan")
    bts_acc = tp / (tp + fn) if tp + fn > 0 else nan
    notbts_acc = tn / (tn + fp) if tn + fp > 0 else nan
    if tp > 0 and tn > 0:
        precision, recall = tp / (tp + fp), tp / (tp + fn)
    else:
        precision = recall = -999.0
    auc = (rank_sum - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg) if n_pos > 0 and n_neg > 0 else nan
    return {"roc_auc": auc, "bal_acc": (bts_acc + notbts_acc) / 2, "bts_acc": bts_acc, "notbts_acc": notbts_acc,
            "alert_precision": precision, "alert_recall": recall,
            "TP": int(tp), "TN": int(tn), "FP": int(fp), "FN": int(fn)}


def run_val(config: dict, model_dir: str, model_filename: str, bts_weight: float, data_base_dir: str = "",
            split: str = "val", device="cuda", precision: str = "bf16"):
    """val.py:31-168 (``run_val(config, model_dir, model_filename, bts_weight, ...)``): instantiate the model the
    config names, load ``model_dir/model_filename`` strictly (a DataParallel ``module.`` prefix is accepted), read
    the split files and run the validation pass.  ``need_triplets`` / ``need_metadata`` follow from the model
    name as in train.py:108-122.  Returns (loss, accuracy, raw_preds, labels) like the reference."""
    import os
    from . import architectures
    from .data import load_split
    from .to_HF import strip_module_prefix
    try:
        model_type = getattr(architectures, config["model_name"])
    except AttributeError:
        raise ValueError(f"Could not find model of name {config['model_name']}") from None
    model = model_type(config, precision=precision)
    state = torch.load(os.path.join(model_dir, model_filename), map_location="cpu")
    model.load_state_dict(strip_module_prefix(state), strict=True)
    model = model.to(device).eval()
    images, metadata, labels, _ = load_split(data_base_dir, config, split)
    return run_val_tensors(model, images, metadata, labels, batch_size=int(config.get("batch_size", 1024)),
                           pos_weight=float(bts_weight), device=device)
