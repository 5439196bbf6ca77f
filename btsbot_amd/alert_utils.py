"""The GPU half of alert -> triplet preprocessing (/root/reference/btsbot/alert_utils.py:110-196).

``make_triplet`` there gunzips and FITS-decodes the three stamps of an alert on the host (astropy),
then masks NaNs, L2-normalises, flags corrupted stamps and pads to 63x63.  Decoding stays host work;
everything after it is one kernel (``btsbot_prep_triplets``) over a whole night's batch, writing the
float32 NCHW tensor the classifier consumes (inference_example.py:62-64) without the float64 NHWC
detour.

    raw, shapes = stack_stamps(list_of_(science, template, difference)_arrays)   # host, after decoding
    triplets, drop = prep_triplets(raw.cuda(), shapes.cuda())
    scores = torch.sigmoid(model(image_input=triplets[~drop], metadata_input=meta[~drop]))
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


def stack_stamps(alerts: Sequence[Sequence[np.ndarray]]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host helper: decoded stamps (each up to 63x63, any float dtype) -> raw [B,3,63,63] float32 with
    every stamp in the top-left corner + shapes int32 [B,3,2]."""
    n = len(alerts)
    raw = np.zeros((n, 3, 63, 63), dtype=np.float32)
    shapes = np.zeros((n, 3, 2), dtype=np.int32)
    for i, trip in enumerate(alerts):
        if len(trip) != 3:
            raise ValueError("every alert needs (science, template, difference) stamps")
        for c, stamp in enumerate(trip):
            h, w = stamp.shape
            if h > 63 or w > 63:
                raise ValueError(f"stamp larger than 63x63: {stamp.shape}")
            raw[i, c, :h, :w] = stamp
            shapes[i, c] = (h, w)
    return torch.from_numpy(raw), torch.from_numpy(shapes)


def prep_triplets(raw: torch.Tensor, shapes: Optional[torch.Tensor] = None,
                  normalize: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(triplets [B,3,63,63] float32, drop [B] bool) on raw's device."""
    if raw.device.type != "cuda":
        raise RuntimeError("btsbot_amd.alert_utils.prep_triplets runs on the GPU; there is no CPU "
                           f"fallback (raw is on {raw.device})")
    if raw.dim() != 4 or tuple(raw.shape[1:]) != (3, 63, 63):
        raise ValueError(f"raw must be [B,3,63,63], got {tuple(raw.shape)}")
    raw = raw.to(torch.float32).contiguous()
    b = raw.shape[0]
    if shapes is not None:
        shapes = shapes.to(device=raw.device, dtype=torch.int32).contiguous()
        if tuple(shapes.shape) != (b, 3, 2):
            raise ValueError(f"shapes must be [{b},3,2], got {tuple(shapes.shape)}")
    out = torch.empty_like(raw)
    drop = torch.zeros(b, dtype=torch.uint8, device=raw.device)
    if b == 0:
        return out, drop.bool()
    with torch.cuda.device(raw.device):
        st = torch.cuda.current_stream(raw.device).cuda_stream
        _lib.check(_lib.lib().btsbot_prep_triplets(
            C.c_void_p(raw.data_ptr()), C.c_void_p(shapes.data_ptr() if shapes is not None else 0),
            C.c_void_p(out.data_ptr()), C.c_void_p(drop.data_ptr()), b, int(normalize),
            C.c_void_p(st)), "btsbot_prep_triplets")
    return out, drop.bool()
