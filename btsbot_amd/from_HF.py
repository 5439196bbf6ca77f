"""Model-hub entry points behind the reference's signatures (/root/reference/btsbot/from_HF.py:16-81).

The contract kept: the accepted ``architecture`` / ``pretrain`` words and the ValueError text for
anything else (from_HF.py:16-29), the hub repository id (``:32-34``), the CWD-relative directory
``models/BTSbot-<arch>-<pretrain>[-metadata]`` (``:37-40``), the two files looked for (``:62-63``), and a
strict ``load_state_dict`` into the class ``train_config.json`` names (``:67-79``).  The returned
module is in train mode, as in the reference.
"""
import json
import os
from typing import NamedTuple

import torch

from . import architectures

device = "cuda" if torch.cuda.is_available() else "cpu"   # MI355X box: always "cuda" (HIP)

_HUB_OWNER = "nabeelr"
_ARCH_TAG = {"convnext": "convnext-pico", "maxvit": "maxvit-tiny"}
_PRETRAIN_TAG = {"imagenet": "in1k", "galaxyzoo": "galaxyzoo", "randinit": "randinit"}
_CHECKPOINT_FILES = ("pytorch_model.bin", "train_config.json")


class _ModelId(NamedTuple):
    arch_tag: str
    multi_modal: bool
    pretrain_tag: str

    @property
    def name(self) -> str:
        suffix = "-metadata" if self.multi_modal else ""
        return f"BTSbot-{self.arch_tag}-{self.pretrain_tag}{suffix}"


def _model_id(architecture: str, multi_modal: bool, pretrain: str) -> _ModelId:
    if architecture not in _ARCH_TAG:
        raise ValueError(f"Invalid architecture: {architecture}")
    if pretrain not in _PRETRAIN_TAG:
        raise ValueError(f"Invalid pre-training regimen: {pretrain}")
    return _ModelId(_ARCH_TAG[architecture], multi_modal, _PRETRAIN_TAG[pretrain])


def validate_model_params(architecture: str, multi_modal: bool, pretrain: str):
    """(architecture tag, multi_modal, pre-training tag) as the hub names them; ValueError otherwise."""
    return tuple(_model_id(architecture, multi_modal, pretrain))


def get_HF_model_link(architecture: str, multi_modal: bool, pretrain: str) -> str:
    return f"{_HUB_OWNER}/{_model_id(architecture, multi_modal, pretrain).name}"


def get_local_model_dir(architecture: str, multi_modal: bool, pretrain: str) -> str:
    return os.path.join("models", _model_id(architecture, multi_modal, pretrain).name)


def download_HF_model(architecture: str, multi_modal: bool, pretrain: str):
    """Snapshot of the hub repository into the local model directory (needs network + huggingface_hub)."""
    repo = get_HF_model_link(architecture, multi_modal, pretrain)
    target = get_local_model_dir(architecture, multi_modal, pretrain)
    try:
        from huggingface_hub import snapshot_download
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("huggingface_hub is required to download BTSbot checkpoints") from e
    print(f"Fetching model from HuggingFace Hub: {repo}")
    os.makedirs(target, exist_ok=True)
    snapshot_download(repo_id=repo, local_dir=target)
    print(f"Model downloaded to {target}")


def _strip_module_prefix(state: dict) -> dict:
    """Checkpoints written from under nn.DataParallel carry a 'module.' prefix (to_onnx.py:31-33)."""
    if state and all(k.startswith("module.") for k in state):
        return {k[len("module."):]: v for k, v in state.items()}
    return state


def load_checkpoint_dir(model_dir: str, dev=None):
    """Model of the class `<model_dir>/train_config.json` names, with `<model_dir>/pytorch_model.bin` loaded."""
    with open(os.path.join(model_dir, "train_config.json")) as f:
        config = json.load(f)
    model = getattr(architectures, config["model_name"])(config).to(dev or device)
    state = torch.load(os.path.join(model_dir, "pytorch_model.bin"), map_location="cpu")
    model.load_state_dict(_strip_module_prefix(state))
    return model


def load_HF_model(architecture: str, multi_modal: bool, pretrain: str):
    model_dir = get_local_model_dir(architecture, multi_modal, pretrain)
    if not all(os.path.isfile(os.path.join(model_dir, f)) for f in _CHECKPOINT_FILES):
        print("Model files not present; downloading model...")
        download_HF_model(architecture, multi_modal, pretrain)
    return load_checkpoint_dir(model_dir)
