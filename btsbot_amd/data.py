"""Training / validation sets resident in HBM, batched and augmented on the device.

Replaces, for data already in memory, what /root/reference/btsbot/train.py:178-209 builds per run:
``FlexibleDataset`` (utils.py:12-41) wrapped in a ``DataLoader(shuffle=True, drop_last=True,
num_workers=6)`` whose per-sample ``__getitem__`` applies ``ToDtype(float32)``,
``RandomHorizontalFlip(0.5)``, ``RandomVerticalFlip(0.5)`` and ``RandomRightAngleRotation``
(utils.py:44-48: a uniform choice of 0/90/180/270 degrees, counter-clockwise like
``transforms.functional.rotate``).  Those transforms are index permutations of the 63x63 cutouts, so
one gather kernel (``btsbot_augment``) produces a shuffled, augmented batch straight from the resident
set: no worker processes, no host copies, no per-sample Python.

    ds = DeviceDataset(images, metadata, labels, batch_size=64, config=config, device="cuda")
    for images_b, meta_b, labels_b in ds:          # one epoch, reshuffled every epoch
        loss = trainer.step(images_b, meta_b, labels_b)

The random stream is torch's device generator (not numpy's / torchvision's), so individual draws
differ from the reference's; the distribution of (permutation, flips, rotation) is the same.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _lib


def augment(images: torch.Tensor, index: Optional[torch.Tensor], ops: Optional[torch.Tensor],
            batch: Optional[int] = None) -> torch.Tensor:
    """dst[b] = rot90^k(vflip?(hflip?(images[index[b]]))), ops[b] = hflip | vflip << 1 | k << 2.
    images [N,3,63,63] f32 on a HIP device; index int64 [B] or None; ops uint8 [B] or None."""
    if images.device.type != "cuda":
        raise RuntimeError("btsbot_amd.data.augment runs on the GPU (libbtsbot_hip.so); there is no "
                           f"CPU fallback (images are on {images.device})")
    if images.dim() != 4 or tuple(images.shape[1:]) != (3, 63, 63) or images.dtype != torch.float32:
        raise ValueError(f"images must be float32 [N,3,63,63], got {images.dtype} {tuple(images.shape)}")
    images = images.contiguous()
    n = batch if batch is not None else (index.numel() if index is not None else images.shape[0])
    if index is not None:
        index = index.to(device=images.device, dtype=torch.int64).contiguous()
        if index.numel() != n:
            raise ValueError("index length != batch")
    elif n > images.shape[0]:
        raise ValueError("batch larger than the image set")
    if ops is not None:
        ops = ops.to(device=images.device, dtype=torch.uint8).contiguous()
        if ops.numel() != n:
            raise ValueError("ops length != batch")
    out = torch.empty(n, 3, 63, 63, dtype=torch.float32, device=images.device)
    if n == 0:
        return out
    with torch.cuda.device(images.device):
        st = torch.cuda.current_stream(images.device).cuda_stream
        _lib.check(_lib.lib().btsbot_augment(
            C.c_void_p(images.data_ptr()), C.c_void_p(index.data_ptr() if index is not None else 0),
            C.c_void_p(ops.data_ptr() if ops is not None else 0), C.c_void_p(out.data_ptr()), n,
            C.c_void_p(st)), "btsbot_augment")
    return out


IMAGE_ONLY_MODELS = ["um_cnn", "SwinV2", "MaxViT", "ConvNeXt"]            # train.py:40-44
METADATA_ONLY_MODELS = ["um_nn"]
MULTIMODAL_MODELS = ["mm_MaxViT", "mm_ConvNeXt", "mm_cnn", "frozen_fusion"]


def model_needs(model_name: str):
    """(need_triplets, need_metadata) of a reference model name (train.py:108-122)."""
    need_triplets = model_name in IMAGE_ONLY_MODELS or model_name in MULTIMODAL_MODELS
    need_metadata = model_name in METADATA_ONLY_MODELS or model_name in MULTIMODAL_MODELS
    if not need_triplets and not need_metadata:
        raise ValueError(f"{model_name} not categorized as image-only/metadata-only/multimodal.")
    return need_triplets, need_metadata


def load_split(data_base_dir: str, config: dict, split: str = "train"):
    """The reference's on-disk split (train.py:133-172 for "train", val.py:82-101 for "val" / "test"):
    ``{data_base_dir}data/{split}_cand_{version}_N{N_max}.csv`` (column ``label`` + the ``metadata_cols`` of the
    config) and ``{split}_triplets_{version}_N{N_max}.npy`` (float64 NHWC [N,63,63,3]) ->
    (triplets [N,3,63,63] float32 NCHW or None, metadata [N,M] float32 or None, labels [N] int64, cand DataFrame).
    The training split drops alerts whose triplet holds a NaN -- from the triplets, the table and the labels alike --
    and refuses NaNs in the metadata columns (ValueError), as the reference does; the other splits are taken as
    they are.  Host tensors: hand them to ``DeviceDataset`` / ``run_val_tensors``, which move them to HBM once."""
    import pandas as pd
    need_triplets, need_metadata = model_needs(config["model_name"])
    version = config["train_data_version"]
    n_str = f"_N{config.get('N_max', 100)}"
    metadata_cols = config.get("metadata_cols", None)
    if need_metadata and metadata_cols is None:
        raise ValueError("Metadata columns not found in config.")
    base = os.path.join(f"{data_base_dir}data", f"{split}_")
    cand = pd.read_csv(f"{base}cand_{version}{n_str}.csv", index_col=None)
    triplets = None
    if need_triplets:
        path = f"{base}triplets_{version}{n_str}.npy"
        if not os.path.exists(path):
            raise FileNotFoundError(f"Triplets file not found for {split}: {path}")
        trip = np.load(path).astype(np.float32)
        if split == "train" and np.any(np.isnan(trip)):
            bad = np.isnan(trip).any(axis=(1, 2, 3))
            trip = trip[~bad]
            cand = cand.loc[~bad].reset_index(drop=True)
        triplets = torch.from_numpy(np.ascontiguousarray(np.transpose(trip, (0, 3, 1, 2))))
    labels = torch.tensor(cand["label"].values, dtype=torch.long)
    metadata = None
    if need_metadata:
        values = cand[metadata_cols].values.astype(np.float32)
        if split == "train" and np.isnan(values).any():
            raise ValueError("NaNs found in metadata columns")
        metadata = torch.from_numpy(values)
    return triplets, metadata, labels, cand


class DeviceDataset:
    """FlexibleDataset + DataLoader for a set that lives in HBM (utils.py:12-41, train.py:178-209,
    val.py:103-115).  ``images`` [N,3,63,63] / ``metadata`` [N,M] may each be None (uni-modal models),
    exactly like the reference's ``need_triplets`` / ``need_metadata``; batches are tuples in the
    reference's order (images, metadata, labels), with the absent member dropped."""

    def __init__(self, images, metadata, labels, batch_size: int, config: Optional[dict] = None,
                 device="cuda", shuffle: bool = True, drop_last: bool = True, augment: bool = True,
                 generator: Optional[torch.Generator] = None, shard: Optional[tuple] = None,
                 check_nan: bool = True):
        config = config or {}
        # shard = (rank, world): every rank holds the whole set and draws the SAME permutation (seed the
        # generators alike); of each global batch it yields its contiguous rows [r*b/w, (r+1)*b/w) -- the
        # reference's DataParallel scatter (train.py:238-240), without moving any alert between GPUs
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        if self.shard is not None:
            r, w = self.shard
            if not (0 <= r < w) or int(batch_size) % w != 0:
                raise ValueError(f"shard {shard}: rank out of range or batch_size {batch_size} not divisible")
            if not drop_last:
                # a ragged last batch can leave a rank with no alerts, and n_global = batch * world is then wrong
                raise ValueError("shard needs drop_last=True")
        self.device = torch.device(device)
        self.images = None if images is None else \
            torch.as_tensor(images).to(self.device, torch.float32).contiguous()
        self.metadata = None if metadata is None else \
            torch.as_tensor(metadata).to(self.device, torch.float32).contiguous()
        self.labels = torch.as_tensor(labels).to(self.device)
        # the reference refuses NaN metadata for the training split only (train.py:170); a validation / test split
        # is taken as it is (val.py:96-99 prints and goes on): those callers pass check_nan=False
        if check_nan and self.metadata is not None and torch.isnan(self.metadata).any():
            raise ValueError("NaNs found in metadata columns")         # train.py:170
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last = shuffle, drop_last
        self.h_flip = augment and bool(config.get("data_aug_h_flip", True))   # train.py:182-184
        self.v_flip = augment and bool(config.get("data_aug_v_flip", True))
        self.rot = augment and bool(config.get("data_aug_rot", True))
        self.generator = generator
        self.num_bts = int((self.labels == 1).sum().item())           # train.py:173-174
        self.num_notbts = int((self.labels == 0).sum().item())

    @property
    def pos_weight(self) -> float:
        """train.py:211: N_neg / N_pos."""
        return self.num_notbts / max(self.num_bts, 1)

    def __len__(self) -> int:
        n = self.labels.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def draw_ops(self, n: int) -> Optional[torch.Tensor]:
        """uint8 [n]: hflip | vflip << 1 | k << 2 with the reference's probabilities."""
        if not (self.h_flip or self.v_flip or self.rot):
            return None
        g, dev = self.generator, self.device
        ops = torch.zeros(n, dtype=torch.uint8, device=dev)
        if self.h_flip:
            ops |= (torch.rand(n, device=dev, generator=g) < 0.5).to(torch.uint8)
        if self.v_flip:
            ops |= (torch.rand(n, device=dev, generator=g) < 0.5).to(torch.uint8) << 1
        if self.rot:
            ops |= torch.randint(0, 4, (n,), device=dev, generator=g, dtype=torch.uint8) << 2
        return ops

    def __iter__(self):
        n = self.labels.shape[0]
        order = torch.randperm(n, device=self.device, generator=self.generator) if self.shuffle \
            else torch.arange(n, device=self.device)
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            if self.shard is not None:
                r, w = self.shard
                per = (idx.numel() + w - 1) // w            # (a ragged last batch, drop_last=False, shards unevenly)
                idx = idx[r * per:(r + 1) * per]
            out = []
            if self.images is not None:
                out.append(augment(self.images, idx, self.draw_ops(idx.numel())))
            if self.metadata is not None:
                out.append(self.metadata[idx])
            out.append(self.labels[idx])
            yield tuple(out)
