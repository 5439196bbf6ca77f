"""Alert -> triplet preprocessing (/root/reference/btsbot/alert_utils.py:110-196).

``make_triplet`` there gunzips and FITS-decodes the three stamps of an alert on the host (astropy),
then masks NaNs, L2-normalises, flags corrupted stamps and pads to 63x63.  Here the decoding is a
dependency-free host step (``decode_stamp``: gzip + the primary HDU of a FITS image -- ZTF cutouts are
single-HDU BITPIX = -32 images) and everything after it is one kernel (``btsbot_prep_triplets``) over a
whole night's batch, writing the float32 NCHW tensor the classifier consumes
(inference_example.py:62-64) without the float64 NHWC detour.

    triplets, drop = make_triplets(alerts, device="cuda")          # alert packets as the reference takes them
    # or, from stamps decoded elsewhere:
    raw, shapes = stack_stamps(list_of_(science, template, difference)_arrays)
    triplets, drop = prep_triplets(raw.cuda(), shapes.cuda())
    scores = torch.sigmoid(model(image_input=triplets[~drop], metadata_input=meta[~drop]))
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

_FITS_DTYPES = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}


def decode_fits_image(buf: bytes) -> np.ndarray:
    """Primary-HDU image of an (uncompressed) FITS file as astropy's ``hdu[0].data`` returns it: shape
    (NAXIS2, NAXIS1) -- NAXIS1 is the fastest axis --, big-endian samples converted to native order,
    BSCALE / BZERO applied when present.  Raises ValueError on anything that is not a 2-D image HDU."""
    cards = {}
    pos, end = 0, None
    while end is None:
        block = buf[pos:pos + 2880]
        if len(block) < 2880:
            raise ValueError("FITS header: END card not found")
        for i in range(0, 2880, 80):
            card = block[i:i + 80].decode("ascii", "replace")
            key = card[:8].strip()
            if key == "END":
                end = pos + 2880
                break
            if card[8:10] == "= ":
                val = card[10:].split("/", 1)[0].strip() if not card[10:].lstrip().startswith("'") \
                    else card[10:].strip().split("'")[1]
                cards[key] = val
        pos += 2880
    try:
        if cards.get("SIMPLE", "T") not in ("T", "t") or int(cards["NAXIS"]) != 2:
            raise ValueError(f"FITS: not a 2-D primary image (NAXIS={cards.get('NAXIS')})")
        bitpix, nx, ny = int(cards["BITPIX"]), int(cards["NAXIS1"]), int(cards["NAXIS2"])
        dt = np.dtype(_FITS_DTYPES[bitpix])
    except KeyError as e:
        raise ValueError(f"FITS header: missing or unsupported {e}") from None
    nbytes = nx * ny * dt.itemsize
    if len(buf) < end + nbytes:
        raise ValueError("FITS: data unit shorter than the header says")
    data = np.frombuffer(buf, dtype=dt, count=nx * ny, offset=end).reshape(ny, nx)
    bscale, bzero = float(cards.get("BSCALE", 1.0)), float(cards.get("BZERO", 0.0))
    if bscale != 1.0 or bzero != 0.0:
        return data.astype(np.float64 if bitpix in (32, 64, -64) else np.float32) * bscale + bzero
    return data.astype(dt.newbyteorder("="))


def decode_stamp(stamp_data) -> np.ndarray:
    """``alert['cutoutScience']['stampData']`` -> the cutout array: gunzip, then the FITS primary image
    (alert_utils.py:139-145 without astropy)."""
    import gzip
    return decode_fits_image(gzip.decompress(bytes(stamp_data)))


def decode_alert(alert) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(science, template, difference) cutouts of one alert packet, in make_triplet's channel order."""
    return tuple(decode_stamp(alert[f"cutout{c}"]["stampData"]) for c in ("Science", "Template", "Difference"))


def make_triplets(alerts, device="cuda", normalize: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """make_triplet (alert_utils.py:110-196) over a batch of alert packets: host decode, then the arithmetic
    on ``device`` in one launch.  Returns (triplets [B,3,63,63] float32 NCHW, drop [B] bool)."""
    raw, shapes = stack_stamps([decode_alert(a) for a in alerts])
    return prep_triplets(raw.to(device), shapes.to(device), normalize)


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
