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


# ---- MaxViT row -----------------------------------------------------
def test_maxvit_definition_matches_published_model_table():
    """No timm here: the restated maxvit_tiny_rw_224 must at least reproduce the two numbers timm's
    model table publishes for it -- 29.1 M parameters, 5.1 GMACs at 224x224."""
    from oracle import maxvit_oracle as MO
    n = MO.count_params("maxvit_tiny_rw_224") + 512 * 1000 + 1000      # + the 1000-class fc
    assert n == 29_075_232 and round(n / 1e6, 1) == 29.1
    assert round(MO.count_macs("maxvit_tiny_rw_224") / 1e9, 1) == 5.1


def test_maxvit_module_form_matches_functional_oracle():
    """oracle/timm_standin.MaxxVitStandIn (nn.Module form, einsum attention, explicit index loops) vs
    oracle/maxvit_oracle (functional form): same timm key set, same features."""
    from oracle import maxvit_oracle as MO, timm_standin as TS
    net = TS.create_model("maxvit_tiny_rw_224.sw_in1k").eval()
    shapes = MO.backbone_param_shapes("maxvit_tiny_rw_224", "")
    keys = [k for k in net.state_dict() if not k.startswith("head.fc")]
    assert keys == list(shapes)
    sd = O.random_state_dict(shapes, seed=5)
    net.load_state_dict(dict(sd, **{"head.fc.weight": torch.zeros(1000, 512),
                                    "head.fc.bias": torch.zeros(1000)}), strict=True)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        a = net.forward_features(x)
        b = MO.forward_features(x, sd, "", "maxvit_tiny_rw_224")
    assert a.shape == (1, 512, 7, 7)
    assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())


def test_maxvit_partitions_are_inverse_pairs_and_rel_index():
    from oracle import maxvit_oracle as MO
    x = torch.arange(2 * 14 * 14 * 3, dtype=torch.float32).view(2, 14, 14, 3)
    assert torch.equal(MO.window_reverse(MO.window_partition(x, 7), 7, 14, 14), x)
    assert torch.equal(MO.grid_reverse(MO.grid_partition(x, 7), 7, 14, 14), x)
    # grid partition: token (gy, gx) of window (iy, ix) is pixel (gy*2 + iy, gx*2 + ix)
    g = MO.grid_partition(x, 7).view(2, 2, 2, 7, 7, 3)
    assert torch.equal(g[1, 1, 0, 3, 5], x[1, 3 * 2 + 1, 5 * 2 + 0])
    idx = MO.rel_pos_index(7)
    assert idx.shape == (49, 49) and idx.min() == 0 and idx.max() == 168
    assert idx[0, 0] == 84 and idx[0, 48] == 0 and idx[48, 0] == 168


@pytest.mark.parametrize("name", ["mm_maxvit", "maxvit", "frozen_fusion_maxvit"])
def test_maxvit_oracle_matches_reference_wrapper_goldens(name):
    """Committed logits of the reference's own MaxViT / mm_MaxViT classes (architectures.py:25-101:
    resize, head surgery, metadata + fusion heads) run around the stand-in backbone."""
    from helpers import MV_CONFIGS, seeded_state_mv
    from oracle import maxvit_oracle as MO
    kind, cfg = MV_CONFIGS[name]
    gold = np.load(os.path.join(GOLD, "ref_logits_maxvit.npz"))
    ex = np.load(os.path.join(GOLD, "example8.npz"))
    sd = seeded_state_mv(kind, cfg, seed=3)
    chk = float(sum(v.double().abs().sum().item() for v in sd.values()))
    assert abs(chk - float(gold[f"{name}/checksum"])) < 1e-6 * chk, "seeded weight stream drifted"
    img = torch.from_numpy(ex["triplets"][[0, 1, 4, 5]])
    meta = torch.from_numpy(ex["metadata"][[0, 1, 4, 5]])
    with torch.no_grad():
        o = MO.forward(kind, sd, cfg, img, meta)
    ref = torch.from_numpy(gold[f"{name}/example4"])
    assert o.shape == ref.shape == (4, 1)
    assert (o - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
