"""Checkpoint hand-over in the reference's on-disk formats (/root/reference/btsbot/to_HF.py:10-43,
train.py best_model.pth / report.json, to_onnx.py:31-33 ``module.`` stripping).

A model trained here and a model trained by stock BTSbot are interchangeable on disk:
``state_dict()`` keys / shapes / order are the reference's, tensors are saved on the CPU.

    save_checkpoint(model, config, model_dir)      # best_model.pth + report.json  (what train.py leaves)
    config = prep_config(model_dir)                # report.json -> train_config.json      to_HF.py:10-24
    prep_model(model_dir, config)                  # best_model.pth -> pytorch_model.bin   to_HF.py:27-43
    btsbot_amd.load_HF_model(...)                  # reads that pair back                  from_HF.py:59-81

Uploading to the hub, model cards and .gitattributes are the reference's release tooling and are not
part of this path.
"""
from __future__ import annotations

import json
import os

import torch

from . import architectures


def strip_module_prefix(state: dict) -> dict:
    """DataParallel checkpoints carry a ``module.`` prefix (train.py:238-240; to_onnx.py:31-33)."""
    if state and all(k.startswith("module.") for k in state):
        return {k[len("module."):]: v for k, v in state.items()}
    return state


def cpu_state_dict(model) -> dict:
    return {k: v.detach().to("cpu").clone() for k, v in model.state_dict().items()}


def save_checkpoint(model, config: dict, model_dir: str, val_summary: dict | None = None) -> str:
    """Write ``best_model.pth`` + ``report.json`` as the reference's training run does."""
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, "best_model.pth")
    torch.save(cpu_state_dict(model), path)
    with open(os.path.join(model_dir, "report.json"), "w") as f:
        json.dump({"train_config": dict(config), "val_summary": dict(val_summary or {})}, f, indent=2)
    return path


def prep_config(model_dir: str) -> dict:
    """to_HF.py:10-24."""
    report_path = os.path.join(model_dir, "report.json")
    if not os.path.exists(report_path):
        raise FileNotFoundError(f"Report file not found: {report_path}")
    with open(report_path, "r") as f:
        config = json.load(f)["train_config"]
    with open(os.path.join(model_dir, "train_config.json"), "w") as f:
        json.dump(config, f, indent=2)
    return config


def prep_model(model_dir: str, config: dict) -> None:
    """to_HF.py:27-43: rebuild the model class from the config, load ``best_model.pth`` strictly,
    save ``pytorch_model.bin``."""
    model_path = os.path.join(model_dir, "best_model.pth")
    if not os.path.exists(model_path):
        raise FileNotFoundError(f"Model file not found: {model_path}")
    model = getattr(architectures, config["model_name"])(config)
    model.load_state_dict(strip_module_prefix(torch.load(model_path, map_location=torch.device("cpu"))))
    torch.save(cpu_state_dict(model), os.path.join(model_dir, "pytorch_model.bin"))
