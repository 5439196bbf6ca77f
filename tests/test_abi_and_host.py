"""CPU-side checks: the C-ABI library loads and exports every symbol of include/btsbot_hip.h,
the parameter table reproduces the reference's state-dict layout, host logic and error behaviour.
No compute call is made (there is no GPU here)."""
import ctypes
import json
import os
import re
import warnings

import pytest
import torch

import btsbot_amd
from btsbot_amd import _lib, from_HF
from helpers import CONFIGS, seeded_state
from oracle import convnext_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(kind, cfg, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return getattr(btsbot_amd, kind)(cfg, **kw)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "btsbot_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(btsbot_[a-z0-9_]+)\(", hdr, re.M))
    assert declared == set(_lib.SYMBOLS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert _lib.lib().btsbot_abi_version() == 1


@pytest.mark.parametrize("name", list(CONFIGS))
def test_state_dict_layout_matches_reference(name):
    kind, cfg = CONFIGS[name]
    m = _build(kind, cfg)
    shapes = O.model_param_shapes(kind, cfg)       # verified == reference wrappers in make_golden
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    # strict load of a reference-shaped state dict, values land in the arena
    ref_sd = seeded_state(kind, cfg, seed=3)
    m.load_state_dict(ref_sd, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k
    assert m.training                                # returned in train mode like the reference


def test_parameters_are_views_of_one_arena():
    kind, cfg = CONFIGS["mm_pico"]
    m = _build(kind, cfg)
    base = m._arena.data_ptr()
    n = m._arena.numel() * 4
    for p in m.parameters():
        assert base <= p.data_ptr() < base + n
    m2 = m.to("cpu").float()
    assert all(m2._arena.data_ptr() <= p.data_ptr() < m2._arena.data_ptr() + n for p in m2.parameters())
    with pytest.raises(TypeError):
        m.half()


def test_forward_on_cpu_fails_loudly():
    kind, cfg = CONFIGS["mm_pico"]
    m = _build(kind, cfg).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(image_input=torch.zeros(2, 3, 63, 63), metadata_input=torch.zeros(2, 25))


def test_bad_configs_raise():
    with pytest.raises(ValueError):
        _build("mm_ConvNeXt", dict(CONFIGS["mm_pico"][1], model_kind="resnet50"))
    with pytest.raises(KeyError):
        _build("mm_ConvNeXt", {k: v for k, v in CONFIGS["mm_pico"][1].items() if k != "comb_fc1_neurons"})
    with pytest.raises(KeyError):                   # like the reference: missing config keys
        btsbot_amd.mm_MaxViT({"pretrained": False})
    with pytest.raises(NotImplementedError):        # legacy VGG-like CNNs are out of scope
        btsbot_amd.mm_cnn({})
    cfg = _lib.make_config("mm_MaxViT", "f32", (2, 2, 6, 2), (64, 128, 256, 512), False, 25, 128,
                           128, 128, 32, 0.1, 0.1)
    with pytest.raises(_lib.BtsbotHipError, match="maxvit_tiny_rw_224"):
        _lib.Handle(cfg)
    with pytest.raises(ValueError):
        _build("um_nn", CONFIGS["um_nn"][1], precision="int8")
    # C side rejects an impossible table directly
    cfg = _lib.make_config("mm_ConvNeXt", "f32", (2, 2, 6, 2), (64, 128, 256, 500), False, 25, 128,
                           128, 128, 32, 0.1, 0.1)
    with pytest.raises(_lib.BtsbotHipError, match="neither convnext_pico nor convnext_nano"):
        _lib.Handle(cfg)


def test_from_hf_name_validation():
    assert from_HF.validate_model_params("convnext", True, "imagenet") == ("convnext-pico", True, "in1k")
    assert from_HF.get_local_model_dir("convnext", True, "galaxyzoo") == \
        os.path.join("models", "BTSbot-convnext-pico-galaxyzoo-metadata")
    assert from_HF.get_HF_model_link("maxvit", False, "randinit") == "nabeelr/BTSbot-maxvit-tiny-randinit"
    with pytest.raises(ValueError, match="Invalid architecture"):
        from_HF.validate_model_params("resnet", False, "imagenet")
    with pytest.raises(ValueError, match="Invalid pre-training regimen"):
        from_HF.validate_model_params("convnext", False, "jft")


def test_load_hf_model_from_local_dir(tmp_path, monkeypatch):
    """load_HF_model reads models/<name>/{train_config.json,pytorch_model.bin} relative to the CWD
    (from_HF.py:37-40,62-79); here with a frozen_fusion checkpoint saved under DataParallel."""
    kind, cfg = CONFIGS["frozen_fusion"]
    sd = seeded_state(kind, cfg, seed=3)
    d = tmp_path / "models" / "BTSbot-convnext-pico-galaxyzoo-metadata"
    d.mkdir(parents=True)
    with open(d / "train_config.json", "w") as f:
        json.dump(cfg, f)
    torch.save({"module." + k: v for k, v in sd.items()}, d / "pytorch_model.bin")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(from_HF, "device", "cpu")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.load_HF_model("convnext", True, "galaxyzoo")
    assert type(m).__name__ == "frozen_fusion"
    assert torch.equal(m.state_dict()["combined_head.5.weight"], sd["combined_head.5.weight"])


def test_frozen_fusion_loads_branch_checkpoints(tmp_path):
    """architectures.py:324-335: branch configs from report.json, weights from best_model.pth."""
    _, icfg = CONFIGS["convnext"]
    _, mcfg = CONFIGS["um_nn"]
    isd = seeded_state("ConvNeXt", icfg, seed=5)
    msd = seeded_state("um_nn", mcfg, seed=6)
    for name, cfg, sd in (("img", icfg, isd), ("meta", mcfg, msd)):
        (tmp_path / name).mkdir()
        with open(tmp_path / name / "report.json", "w") as f:
            json.dump({"train_config": cfg}, f)
        torch.save(sd, tmp_path / name / "best_model.pth")
    cfg = dict(model_name="frozen_fusion", image_model_dir=str(tmp_path / "img"),
               meta_model_dir=str(tmp_path / "meta"), comb_fc1_neurons=64, comb_fc2_neurons=16,
               comb_dropout=0.1)
    m = _build("frozen_fusion", cfg)
    out = m.state_dict()
    assert torch.equal(out["image_branch.convnext.stages.2.blocks.3.mlp.fc1.weight"],
                       isd["convnext.stages.2.blocks.3.mlp.fc1.weight"])
    assert torch.equal(out["image_branch.convnext.head.1.weight"], isd["convnext.head.1.weight"])
    assert torch.equal(out["meta_branch.network.4.bias"], msd["network.4.bias"])
    assert "meta_branch.network.6.weight" not in out       # head removed (:299-303)


def test_weight_version_tracking():
    kind, cfg = CONFIGS["um_nn"]
    m = _build(kind, cfg)
    v0 = m._version()
    with torch.no_grad():
        next(m.parameters()).add_(1.0)
    assert m._version() != v0


@pytest.mark.parametrize("name", ["mm_maxvit", "maxvit", "frozen_fusion_maxvit"])
def test_maxvit_state_dict_layout_matches_reference(name):
    """timm MaxxVit key names / shapes / order (verified == the reference wrappers around the stand-in
    in make_golden.py), BatchNorm buffers included; strict load round-trips."""
    from helpers import MV_CONFIGS, seeded_state_mv
    from oracle import maxvit_oracle as MO
    kind, cfg = MV_CONFIGS[name]
    m = _build(kind, cfg)
    shapes = MO.model_param_shapes(kind, cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    ref_sd = seeded_state_mv(kind, cfg, seed=3)
    m.load_state_dict(ref_sd, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k
    assert m.image_size == 224 and m.training
    bad = dict(cfg, model_kind="maxvit_base_tf_384.in1k") if kind != "frozen_fusion" else \
        dict(cfg, image_model_config=dict(cfg["image_model_config"], model_kind="maxvit_base_tf_384.in1k"))
    with pytest.raises(ValueError):
        _build(kind, bad)


def test_load_hf_model_maxvit_metadata_from_local_dir(tmp_path, monkeypatch):
    """``load_HF_model("maxvit", True, ...)`` is what the reference's README tells users to call: the
    "-metadata" checkpoints are frozen_fusion models with a MaxViT image branch (from_HF.py:59-81,
    architectures.py:304-308)."""
    from helpers import MV_CONFIGS, seeded_state_mv
    kind, cfg = MV_CONFIGS["frozen_fusion_maxvit"]
    sd = seeded_state_mv(kind, cfg, seed=3)
    mdir = tmp_path / "models" / "BTSbot-maxvit-tiny-randinit-metadata"
    mdir.mkdir(parents=True)
    torch.save({"module." + k: v for k, v in sd.items()}, mdir / "pytorch_model.bin")
    with open(mdir / "train_config.json", "w") as f:
        json.dump(cfg, f)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(from_HF, "device", "cpu")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.load_HF_model("maxvit", True, "randinit")
    assert type(m).__name__ == "frozen_fusion" and m.image_size == 224
    got = m.state_dict()
    assert list(got) == list(sd)
    for k in ("image_branch.maxvit.stem.conv1.weight", "image_branch.maxvit.norm.bias",
              "meta_branch.network.4.weight", "combined_head.5.bias",
              "image_branch.maxvit.stages.2.blocks.4.attn_grid.attn.rel_pos.relative_position_bias_table"):
        assert torch.equal(got[k], sd[k]), k


def test_maxvit_train_mode_gate_is_host_logic():
    """Which training-mode calls of the MaxViT wirings are served is decided on the host before any kernel runs
    (architectures._check_train_supported): the image branch trains as a whole (every BatchNorm2d holder in train mode:
    batch statistics + the backward of every layer) or is frozen and in eval mode (heads train over fixed features);
    the two mixed regimes are refused.  ``model.train()`` picks the regime from the branch's requires_grad flags.
    No GPU needed."""
    from helpers import MV_CONFIGS
    kind, cfg = MV_CONFIGS["mm_maxvit"]
    m = _build(kind, cfg).train()
    bn = m._image_bn_modules()
    n_bn = sum(1 for k in m.state_dict() if k.startswith("maxvit_backbone.") and k.endswith("running_mean"))
    assert len(bn) == n_bn and n_bn > 30
    # every parameter trainable: model.train() is the reference's (batch statistics everywhere) and the branch trains
    assert m.training and all(b.training for b in bn)
    m._check_train_supported(True)
    with pytest.raises(NotImplementedError, match="FROZEN"):
        m._check_train_supported(False)                   # batch statistics without the branch's backward: refused
    # an eval-mode branch: heads train over its fixed features, its own gradients are refused
    m.maxvit_backbone.eval()
    assert m.training and not any(b.training for b in bn)
    m._check_train_supported(False)
    with pytest.raises(NotImplementedError, match="eval"):
        m._check_train_supported(True)
    # a FROZEN branch: model.train() (e.g. after a validation pass, train.py:332-340) keeps its BatchNorm2d holders
    # in eval mode, so the next step is served
    m.maxvit_backbone.requires_grad_(False)
    m.eval()
    m.train()
    assert m.training and not any(b.training for b in bn)
    m._check_train_supported(False)
    # ... and thawing it brings batch statistics back
    m.maxvit_backbone.requires_grad_(True)
    m.train()
    assert all(b.training for b in bn)
    m._check_train_supported(True)
    # the ConvNeXt wirings are not gated at all
    from helpers import CONFIGS
    kind2, cfg2 = CONFIGS["mm_pico"]
    _build(kind2, cfg2).train()._check_train_supported(True)


def test_score_stream_host_checks():
    """ScoreStream (btsbot_amd/pipeline.py) refuses what it cannot run: depth < 1, a training-mode model, a model that
    is not on the GPU (no CPU fallback)."""
    import warnings
    import btsbot_amd
    from helpers import CONFIGS
    kind, cfg = CONFIGS["um_nn"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = getattr(btsbot_amd, kind)(cfg)
    with pytest.raises(ValueError):
        btsbot_amd.ScoreStream(m.eval(), depth=0)
    with pytest.raises(RuntimeError, match="eval"):
        btsbot_amd.ScoreStream(m.train(), depth=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        btsbot_amd.ScoreStream(m.eval(), depth=2)
    assert m._init_config["meta_fc1_neurons"] == cfg["meta_fc1_neurons"]
