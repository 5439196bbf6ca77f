"""Shared test configuration: model configs and seeded weights (oracle side)."""
import warnings

import torch

from btsbot_amd.synthetic import METADATA_COLS
from oracle import convnext_oracle as O

MM_PICO = dict(model_name="mm_ConvNeXt", model_kind="convnext_pico.d1_in1k", pretrained=False,
               train_data_version="v11", metadata_cols=METADATA_COLS, meta_fc1_neurons=128,
               meta_dropout=0.25, meta_fc2_neurons=128, comb_fc1_neurons=128,
               comb_fc2_neurons=32, comb_dropout=0.2)
MM_NANO_LS = dict(MM_PICO, model_kind="convnext_nano.d1h_in1k", train_data_version="v11_LS")
IMG_PICO = dict(model_name="ConvNeXt", model_kind="convnext_pico.d1_in1k", pretrained=False,
                fc1_neurons=64, fc2_neurons=16, dropout=0.1)
META = dict(model_name="um_nn", metadata_cols=METADATA_COLS, meta_fc1_neurons=128,
            meta_dropout=0.25, meta_fc2_neurons=64)
FUSION = dict(model_name="frozen_fusion", image_model_dir="unused", meta_model_dir="unused",
              image_model_config=IMG_PICO, meta_model_config=META, skip_load_state=True,
              comb_fc1_neurons=64, comb_fc2_neurons=16, comb_dropout=0.1)

MM_MAXVIT = dict(MM_PICO, model_name="mm_MaxViT", model_kind="maxvit_tiny_rw_224.sw_in1k")
IMG_MAXVIT = dict(model_name="MaxViT", model_kind="maxvit_tiny_rw_224.sw_in1k", pretrained=False,
                  fc1_neurons=64, fc2_neurons=16, dropout=0.1)
FUSION_MAXVIT = dict(model_name="frozen_fusion", image_model_dir="unused", meta_model_dir="unused",
                     image_model_config=IMG_MAXVIT, meta_model_config=META, skip_load_state=True,
                     comb_fc1_neurons=64, comb_fc2_neurons=16, comb_dropout=0.1)
MV_CONFIGS = {"mm_maxvit": ("mm_MaxViT", MM_MAXVIT), "maxvit": ("MaxViT", IMG_MAXVIT),
              "frozen_fusion_maxvit": ("frozen_fusion", FUSION_MAXVIT)}

CONFIGS = {"mm_pico": ("mm_ConvNeXt", MM_PICO), "mm_nano_ls": ("mm_ConvNeXt", MM_NANO_LS),
           "convnext": ("ConvNeXt", IMG_PICO), "um_nn": ("um_nn", META),
           "frozen_fusion": ("frozen_fusion", FUSION)}


def seeded_state(kind: str, config: dict, seed: int, gamma: float = 1.0):
    return O.random_state_dict(O.model_param_shapes(kind, config), seed, gamma)


def seeded_state_mv(kind: str, config: dict, seed: int):
    from oracle import maxvit_oracle as MO
    return O.random_state_dict(MO.model_param_shapes(kind, config), seed)


def build_model(kind: str, config: dict, sd: dict, device, precision="f32"):
    import btsbot_amd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = getattr(btsbot_amd, kind)(config, precision=precision)
    m.load_state_dict(sd)
    return m.to(device).eval()


def run_model(kind, m, img, meta):
    with torch.no_grad():
        if kind in ("mm_ConvNeXt", "frozen_fusion", "mm_MaxViT"):   # (frozen_fusion: either image branch)
            return m(image_input=img, metadata_input=meta)
        if kind in ("ConvNeXt", "MaxViT"):
            return m(input_data=img)
        return m(input_data=meta)
