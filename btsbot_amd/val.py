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
                       drop_last=False, augment=False)
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
