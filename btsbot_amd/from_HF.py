"""Model-hub entry points with the reference's signatures (/root/reference/btsbot/from_HF.py).

``load_HF_model(architecture, multi_modal, pretrain)`` looks in the CWD-relative
``models/BTSbot-<arch>-<pretrain>[-metadata]/`` directory for ``pytorch_model.bin`` +
``train_config.json`` (from_HF.py:37-40,62-63), builds the class named by
``config["model_name"]`` from :mod:`btsbot_amd.architectures` and loads the state dict strictly
(from_HF.py:67-79).  The returned module is in train mode, as in the reference.
"""
import json
import os

import torch

from . import architectures

device = "cuda" if torch.cuda.is_available() else "cpu"   # from_HF.py:7-13 (no mps on MI355X)


def validate_model_params(architecture: str, multi_modal: bool, pretrain: str):
    """from_HF.py:16-29 -- same names, same ValueErrors."""
    if architecture == "convnext":
        architecture = "convnext-pico"
    elif architecture == "maxvit":
        architecture = "maxvit-tiny"
    else:
        raise ValueError(f"Invalid architecture: {architecture}")

    if pretrain == "imagenet":
        pretrain = "in1k"
    elif pretrain not in ["galaxyzoo", "randinit"]:
        raise ValueError(f"Invalid pre-training regimen: {pretrain}")

    return architecture, multi_modal, pretrain


def get_HF_model_link(architecture: str, multi_modal: bool, pretrain: str) -> str:
    architecture, multi_modal, pretrain = validate_model_params(architecture, multi_modal, pretrain)
    return "nabeelr/BTSbot-" + architecture + "-" + pretrain + ("-metadata" if multi_modal else "")


def get_local_model_dir(architecture: str, multi_modal: bool, pretrain: str) -> str:
    architecture, multi_modal, pretrain = validate_model_params(architecture, multi_modal, pretrain)
    model_name = "BTSbot-" + architecture + "-" + pretrain + ("-metadata" if multi_modal else "")
    return os.path.join("models", model_name)


def download_HF_model(architecture: str, multi_modal: bool, pretrain: str):
    """from_HF.py:43-56.  Needs network access and huggingface_hub; raises a clear error otherwise."""
    HF_link = get_HF_model_link(architecture, multi_modal, pretrain)
    model_dir = os.path.join("models", HF_link.split("/")[-1])
    try:
        from huggingface_hub import snapshot_download
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("huggingface_hub is required to download BTSbot checkpoints") from e
    print(f"Fetching model from HuggingFace Hub: {HF_link}")
    os.makedirs(model_dir, exist_ok=True)
    snapshot_download(repo_id=HF_link, local_dir=model_dir)
    print(f"Model downloaded to {model_dir}")


def load_HF_model(architecture: str, multi_modal: bool, pretrain: str):
    model_dir = get_local_model_dir(architecture, multi_modal, pretrain)

    required_files = ["pytorch_model.bin", "train_config.json"]
    if not all(os.path.isfile(os.path.join(model_dir, f)) for f in required_files):
        print("Model files not present; downloading model...")
        download_HF_model(architecture, multi_modal, pretrain)

    with open(os.path.join(model_dir, "train_config.json"), "r") as f:
        config = json.load(f)

    model_type = getattr(architectures, config["model_name"])
    model = model_type(config).to(device)
    state = torch.load(os.path.join(model_dir, "pytorch_model.bin"),
                       map_location=torch.device("cpu"))
    if state and next(iter(state.keys())).startswith("module."):     # to_onnx.py:31-33
        state = {k[len("module."):]: v for k, v in state.items()}
    model.load_state_dict(state)
    return model
