#!/usr/bin/env python3
"""A whole training run on the MI355X-native package, in the shapes the reference's train.py works with.

    python examples/train_example.py [--model mm_ConvNeXt|um_nn|ConvNeXt] [--alerts 4096] [--epochs 3]
    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 2 examples/train_example.py   # one rank per GPU

It writes a synthetic split in the reference's on-disk layout (``data/{train,val}_cand_v11_N100.csv`` +
``..._triplets_v11_N100.npy``: L2-normalised cutouts with a central source, the 25 metadata columns, labels
that depend on both), then runs ``btsbot_amd.train.run_training`` with a config of the reference's keys
(train.py:75-440: seeds, AdamW + warm-up / cosine schedule, BCE with pos_weight, latest / best checkpoints,
early stopping, report.json), turns the result into the Hugging-Face pair with ``to_HF.prep_config`` /
``prep_model`` and scores the validation split again through ``val.run_val``.  There is no network here, so
``pretrained`` is False and the data is synthetic; point ``--data-base-dir`` at a directory that already holds
real split files to train on those instead.
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from btsbot_amd import to_HF  # noqa: E402
from btsbot_amd.synthetic import METADATA_COLS, synthetic_batch  # noqa: E402
from btsbot_amd.train import run_training  # noqa: E402
from btsbot_amd.val import alert_summary, run_val  # noqa: E402


def write_split(base, split, n, seed):
    img, meta, _ = synthetic_batch(n, seed=seed)
    # a label both modalities carry: brighter-than-median central pixel of the difference cutout XOR-free with
    # a metadata threshold, so image-only, metadata-only and multi-modal models all have something to learn
    bright = img[:, 2, 31, 31] > img[:, 2, 31, 31].median()
    label = (bright & (meta[:, 5] > meta[:, 5].quantile(0.3))).long().numpy()
    os.makedirs(os.path.join(base, "data"), exist_ok=True)
    np.save(os.path.join(base, "data", f"{split}_triplets_v11_N100.npy"),
            img.permute(0, 2, 3, 1).contiguous().numpy().astype(np.float64))          # NHWC float64, as on disk
    df = pd.DataFrame(meta.numpy(), columns=METADATA_COLS)
    df["label"] = label
    df.to_csv(os.path.join(base, "data", f"{split}_cand_v11_N100.csv"), index=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mm_ConvNeXt", choices=["mm_ConvNeXt", "ConvNeXt", "um_nn"])
    ap.add_argument("--alerts", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=["f32", "bf16", "f16"])
    ap.add_argument("--data-base-dir", default=None)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    rank = int(os.environ.get("RANK", "0"))

    base = args.data_base_dir
    tmp = None
    if base is None:
        tmp = tempfile.mkdtemp(prefix="btsbot_amd_example_") if rank == 0 else None
        if world > 1:
            box = [tmp]
            torch.distributed.broadcast_object_list(box, src=0)
            tmp = box[0]
        base = tmp + "/"
        if rank == 0:
            write_split(base, "train", args.alerts, seed=1)
            write_split(base, "val", max(args.alerts // 4, 64), seed=2)
        if world > 1:
            torch.distributed.barrier()

    config = dict(model_name=args.model, model_kind="convnext_pico.d1_in1k", pretrained=False,
                  train_data_version="v11", N_max=100, metadata_cols=METADATA_COLS,
                  meta_fc1_neurons=128, meta_dropout=0.25, meta_fc2_neurons=128, comb_fc1_neurons=128,
                  comb_fc2_neurons=32, comb_dropout=0.2, fc1_neurons=128, fc2_neurons=32, dropout=0.2,
                  epochs=args.epochs, batch_size=args.batch_size, learning_rate=1e-3, warmup_epochs=1,
                  beta_1=0.9, beta_2=0.999, patience=5, random_seed=2)
    hist, model_dir = run_training(config, data_base_dir=base, run_name="example", device=dev,
                                   precision=args.precision, models_root=os.path.join(base, "models"))
    if rank == 0:
        for e, (tl, ta, vl, va) in enumerate(zip(hist["train_loss"], hist["train_accuracy"], hist["val_loss"],
                                                 hist["val_accuracy"])):
            print(f"epoch {e + 1}: train loss {tl:.4f} acc {ta:.3f} | val loss {vl:.4f} acc {va:.3f}")
        print("val_summary:", {k: round(v, 4) if isinstance(v, float) else v for k, v in hist["val_summary"].items()})
        cfg = to_HF.prep_config(model_dir)                  # report.json -> train_config.json
        to_HF.prep_model(model_dir, cfg)                    # best_model.pth -> pytorch_model.bin
        train_lab = pd.read_csv(os.path.join(base, "data", "train_cand_v11_N100.csv"))["label"].values
        pw = float((train_lab == 0).sum() / max((train_lab == 1).sum(), 1))
        vl, va, raw, lab = run_val(config, model_dir, "best_model.pth", pw, data_base_dir=base, split="val",
                                   device=dev, precision=args.precision)
        print(f"run_val on the best checkpoint: loss {vl:.4f} accuracy {va:.3f} ROC AUC "
              f"{alert_summary(raw, lab)['roc_auc']:.3f}; files in {model_dir}: {sorted(os.listdir(model_dir))}")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
