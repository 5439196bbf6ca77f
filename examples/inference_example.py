#!/usr/bin/env python3
"""Counterpart of /root/reference/btsbot/inference_example.py on the MI355X-native package.

Same command line (``--architecture {convnext,maxvit} [--pretrain ...] [--multi_modal]``), same data
preparation (the 25 metadata columns in the reference's order, float32 cast, NHWC -> NCHW), same
first-batch-of-64 forward and the same two printed lines (rounded scores, labels).  Differences: the
model comes from ``btsbot_amd.load_HF_model`` (reads ``models/BTSbot-.../`` exactly like the
reference; no network here to download), the bundled example files are looked up in ``--data-dir``
(default ``example_data``, i.e. run it from a BTSbot checkout's ``btsbot/`` directory as the reference
is), the batch is fed from a ``DeviceDataset`` instead of DataLoader worker processes, and
``--random-weights`` builds the architecture with seeded random weights when no checkpoint directory
exists (smoke runs).
"""
import argparse
import os
import sys
import warnings

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import btsbot_amd as btsbot  # noqa: E402
from btsbot_amd.data import DeviceDataset  # noqa: E402

METADATA_COLS = [  # inference_example.py:53-58
    "sgscore1", "distpsnr1", "sgscore2", "distpsnr2", "fwhm", "magpsf",
    "sigmapsf", "chipsf", "ra", "dec", "diffmaglim", "ndethist", "nmtchps",
    "age", "days_since_peak", "days_to_peak", "peakmag_so_far", "new_drb",
    "ncovhist", "nnotdet", "chinr", "sharpnr", "scorr", "sky", "maxmag_so_far"]


def parse_args():
    p = argparse.ArgumentParser(description="Run a BTSbot model on the bundled example alerts (MI355X)")
    p.add_argument("--architecture", type=str, required=True, choices=["convnext", "maxvit"])
    p.add_argument("--pretrain", type=str, default="galaxyzoo", choices=["imagenet", "galaxyzoo", "randinit"])
    p.add_argument("--multi_modal", action="store_true")
    p.add_argument("--data-dir", type=str, default="example_data")
    p.add_argument("--precision", type=str, default=None, choices=[None, "f32", "bf16", "f16", "fp8"])
    p.add_argument("--all-batches", action="store_true",
                   help="score the whole file, not just its first batch, through btsbot.ScoreStream (two batches in flight)")
    p.add_argument("--random-weights", action="store_true",
                   help="no checkpoint: seeded random weights of the chosen architecture")
    return p.parse_args()


def prepare_inputs(cand: pd.DataFrame, triplets: np.ndarray, multi_modal: bool):
    """inference_example.py:47-64: labels int64, metadata float32 in METADATA_COLS order (multi-modal
    only), triplets float32 NCHW contiguous."""
    labels = torch.tensor(cand["label"].values, dtype=torch.long)
    metadata = torch.tensor(cand[METADATA_COLS].values.astype(np.float32)) if multi_modal else None
    trip = np.transpose(np.asarray(triplets).astype(np.float32), (0, 3, 1, 2))
    return torch.from_numpy(np.ascontiguousarray(trip)), metadata, labels


def run_inference(model, multi_modal: bool, data_dir: str, device="cuda"):
    cand = pd.read_csv(os.path.join(data_dir, "usage_candidates.csv"), index_col=None)
    triplets = np.load(os.path.join(data_dir, "usage_triplets.npy"), mmap_mode="r")
    images, metadata, labels = prepare_inputs(cand, triplets, multi_modal)
    ds = DeviceDataset(images, metadata, labels, batch_size=64, device=device, shuffle=False,
                       drop_last=False, augment=False)
    model = model.to(device).eval()
    with torch.no_grad():
        batch = next(iter(ds))
        if multi_modal:
            images_batch, meta_batch, labels_batch = batch
            logits = model(image_input=images_batch, metadata_input=meta_batch)
        else:
            images_batch, labels_batch = batch
            logits = model(input_data=images_batch)
        raw_preds = torch.sigmoid(logits).round().squeeze().cpu().numpy().astype(int)
    print(raw_preds)
    print(labels_batch.cpu().numpy())
    return raw_preds, labels_batch.cpu().numpy()


def run_all_batches(model, multi_modal: bool, data_dir: str, device="cuda"):
    """The scoring loop over every batch of the file: what val.py:103-157 does with one model(...) call per batch, here
    with consecutive batches on alternating HIP streams (btsbot.ScoreStream)."""
    cand = pd.read_csv(os.path.join(data_dir, "usage_candidates.csv"), index_col=None)
    triplets = np.load(os.path.join(data_dir, "usage_triplets.npy"), mmap_mode="r")
    images, metadata, labels = prepare_inputs(cand, triplets, multi_modal)
    ds = DeviceDataset(images, metadata, labels, batch_size=64, device=device, shuffle=False,
                       drop_last=False, augment=False, check_nan=False)
    scorer = btsbot.ScoreStream(model.to(device).eval(), depth=2)
    scores = [torch.sigmoid(z).squeeze(1) for z in scorer.map(batch[:-1] for batch in ds)]
    scores = torch.cat(scores).cpu().numpy()
    print(f"{len(scores)} alerts scored, {int((scores >= 0.5).sum())} above 0.5")
    return scores


def random_model(architecture: str, multi_modal: bool, precision):
    cfg = dict(pretrained=False, train_data_version="v11", metadata_cols=METADATA_COLS,
               meta_fc1_neurons=128, meta_fc2_neurons=128, meta_dropout=0.25, comb_fc1_neurons=128,
               comb_fc2_neurons=32, comb_dropout=0.2, fc1_neurons=128, fc2_neurons=32, dropout=0.2)
    cfg["model_kind"] = "convnext_pico.d1_in1k" if architecture == "convnext" else "maxvit_tiny_rw_224.sw_in1k"
    name = {("convnext", True): "mm_ConvNeXt", ("convnext", False): "ConvNeXt",
            ("maxvit", True): "mm_MaxViT", ("maxvit", False): "MaxViT"}[(architecture, multi_modal)]
    torch.manual_seed(2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return getattr(btsbot, name)(cfg, precision=precision)


if __name__ == "__main__":
    args = parse_args()
    if args.random_weights:
        model = random_model(args.architecture, args.multi_modal, args.precision)
    else:
        model = btsbot.load_HF_model(args.architecture, args.multi_modal, args.pretrain)
        if args.precision:
            model.set_precision(args.precision)
    if args.all_batches:
        run_all_batches(model, args.multi_modal, args.data_dir)
    else:
        run_inference(model, args.multi_modal, args.data_dir)
