"""The oracle is only as good as its pins (CPU, no GPU, no /root/reference needed):
 (i)   functional oracle == transformers.ConvNextModel (independent ConvNeXt implementation)
 (ii)  functional oracle == committed logits of the REFERENCE's own wrapper classes
       (tests/golden/ref_logits.npz, produced by tests/golden/make_golden.py in the build container)
 (iii) closed-form BCE gradient == autograd; adamw_step == torch.optim.AdamW trajectory;
       lr_sequence == committed torch SequentialLR sequences
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import CONFIGS, seeded_state
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hf_key(k):
    k = k.replace("stem.0", "embeddings.patch_embeddings").replace("stem.1", "embeddings.layernorm")
    k = k.replace("stages.", "encoder.stages.").replace("downsample.", "downsampling_layer.")
    k = k.replace("blocks.", "layers.").replace("conv_dw", "dwconv").replace(".norm.", ".layernorm.")
    k = k.replace("mlp.fc1", "pwconv1").replace("mlp.fc2", "pwconv2")
    k = k.replace("gamma", "layer_scale_parameter").replace("head.layernorm.", "layernorm.")
    return k


@pytest.mark.parametrize("arch", ["convnext_pico", "convnext_nano"])
def test_backbone_matches_transformers_convnext(arch):
    from transformers import ConvNextConfig, ConvNextModel
    t = O.ARCHS[arch]
    hf = ConvNextModel(ConvNextConfig(num_channels=3, patch_size=4, hidden_sizes=list(t["dims"]),
                                      depths=list(t["depths"]), layer_scale_init_value=1.0)).eval()
    sd = O.random_state_dict(O.backbone_param_shapes(arch, "", True, "head.norm."), seed=1)
    hsd = {}
    for k, v in sd.items():
        k2 = _hf_key(k)
        hsd[k2] = v.flatten(1) if ("pwconv" in k2 and k2.endswith("weight")) else v
    hf.load_state_dict(hsd, strict=True)
    img, _, _ = synthetic_batch(3, seed=5)
    with torch.no_grad():
        out = hf(img)
        x = O.forward_features(img, sd, "", arch)
        pooled = O.pooled_head(x, sd["head.norm.weight"], sd["head.norm.bias"])
    assert x.shape == (3, t["dims"][-1], 1, 1)            # 63 -> 15 -> 7 -> 3 -> 1
    assert (out.last_hidden_state - x).abs().max() < 5e-5
    assert (out.pooler_output - pooled).abs().max() < 5e-5


@pytest.mark.parametrize("name", list(CONFIGS))
def test_oracle_matches_reference_wrapper_goldens(name):
    kind, cfg = CONFIGS[name]
    gold = np.load(os.path.join(GOLD, "ref_logits.npz"))
    ex = np.load(os.path.join(GOLD, "example8.npz"))
    sd = seeded_state(kind, cfg, seed=3)
    chk = float(sum(v.double().abs().sum().item() for v in sd.values()))
    assert abs(chk - float(gold[f"{name}/checksum"])) < 1e-6 * chk, "seeded weight stream drifted"
    simg, smeta, _ = synthetic_batch(6, seed=2)
    assert abs(simg.double().abs().sum().item() - float(gold["synthetic6/img_checksum"])) < 1e-6
    for tag, img, meta in (("example8", torch.from_numpy(ex["triplets"]), torch.from_numpy(ex["metadata"])),
                           ("synthetic6", simg, smeta)):
        with torch.no_grad():
            o = O.forward(kind, sd, cfg, img, meta)
        ref = torch.from_numpy(gold[f"{name}/{tag}"])
        assert o.shape == ref.shape == (img.shape[0], 1)
        scale = max(1.0, ref.abs().max().item())
        assert (o - ref).abs().max().item() < 5e-5 * scale


def test_example8_fixture_shape_and_normalisation():
    ex = np.load(os.path.join(GOLD, "example8.npz"))
    assert ex["triplets"].shape == (8, 3, 63, 63) and ex["triplets"].dtype == np.float32
    assert ex["metadata"].shape == (8, 25)
    assert ex["labels"].tolist() == [1, 1, 1, 1, 0, 0, 0, 0]
    norms = np.sqrt((ex["triplets"].astype(np.float64) ** 2).sum(axis=(2, 3)))
    assert np.allclose(norms, 1.0, atol=1e-4)             # alert_utils.py:162-164 L2 normalisation
    assert (ex["expected_scores"][:4] > 0.98).all() and (ex["expected_scores"][4:] < 2e-3).all()


def test_bce_value_and_gradient():
    g = np.load(os.path.join(GOLD, "adamw_bce.npz"))
    z, y, pw = torch.from_numpy(g["z"]), torch.from_numpy(g["y"]), float(g["pos_weight"])
    assert abs(O.bce_with_logits(z, y, pw).item() - float(g["loss"])) < 1e-6
    assert np.allclose(O.bce_grad(z, y, pw).numpy(), g["dz"], atol=1e-7)
    # and against autograd directly
    zz = z.clone().requires_grad_(True)
    O.bce_with_logits(zz, y, pw).backward()
    assert torch.allclose(zz.grad, O.bce_grad(z, y, pw), atol=1e-7)


def test_adamw_trajectory():
    g = np.load(os.path.join(GOLD, "adamw_bce.npz"))
    p = torch.from_numpy(g["p0"]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(3):
        O.adamw_step(p, torch.from_numpy(g["grads"][s]), m, v, s + 1, 1e-4, 0.99, 0.99)
        assert np.allclose(p.numpy(), g["traj"][s], rtol=1e-6, atol=1e-7)


def test_lr_sequences():
    with open(os.path.join(GOLD, "lr_sequences.json")) as f:
        gold = json.load(f)
    for key, seq in gold.items():
        w, e = map(int, key.split(","))
        assert np.allclose(O.lr_sequence(1e-4, e, w), seq, rtol=1e-9)
    # warmup 0 leaves the LR stuck at 0.01*lr with torch 2.10
    assert np.allclose(gold["2,8"][:3], [1e-6, 5.05e-5, 1e-4], rtol=1e-6)


def test_grads_match_autograd_through_functional_oracle():
    """The oracle's forward is differentiable torch code: d(loss)/d(param) via autograd is the
    gradient oracle for the HIP backward; here we only check it is well-formed and deterministic."""
    kind, cfg = CONFIGS["mm_pico"]
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
          for k, v in seeded_state(kind, cfg, seed=3).items()}
    img, meta, lab = synthetic_batch(4, seed=2)
    logits = O.forward(kind, sd, cfg, img, meta)
    loss = O.bce_with_logits(logits, lab.float().unsqueeze(1), 1.0)
    loss.backward()
    gn = {k: v.grad.norm().item() for k, v in sd.items() if getattr(v, "grad", None) is not None}
    assert len(gn) == 136 and all(np.isfinite(x) for x in gn.values())
    assert gn["convnext_backbone.stem.0.weight"] > 0
