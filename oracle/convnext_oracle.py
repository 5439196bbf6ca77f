"""fp32 CPU restatement of BTSbot's ConvNeXt multi-modal classifier path.

TEST INFRASTRUCTURE (checker + CPU baseline) -- never imported by the product package.

What it restates (all paths relative to /root/reference):
  * timm ConvNeXt (pico / nano, conv_mlp=True, patch stem) reached through
    ``timm.create_model`` at btsbot/architectures.py:108,132.  timm (>=0.9.0, floor only,
    pyproject.toml:43) is NOT in the reference tree and not installed here; the algorithm
    restated is the published ConvNeXt-v1 block:
        x + gamma * fc2(GELU_erf(fc1(LN_C(dwconv7x7_p3(x)))))            (block)
        LN_C(conv4x4_s4(img))                                            (stem)
        conv2x2_s2(LN_C(x))                                              (downsample)
    with LN eps 1e-6 (biased variance) and state-dict names as timm 1.0 writes them.
  * the wrappers btsbot/architectures.py:104-171 (ConvNeXt, mm_ConvNeXt),
    :277-293 (um_nn), :296-372 (frozen_fusion).
  * loss / optimiser / schedule: btsbot/train.py:211-212 (BCEWithLogits, pos_weight),
    :242-246 (AdamW betas, torch defaults eps 1e-8 wd 1e-2), :249-260 (warm-up + cosine).

How it is pinned (oracle/README.md; tests/test_oracle_pins.py; tests/golden/make_golden.py):
  (i)  against ``transformers.ConvNextModel`` (independent implementation of the same
       published architecture) with mapped weights -- runs everywhere, no reference needed;
  (ii) against the reference's OWN wrapper classes imported from /root/reference with a
       stand-in ``timm`` (oracle/timm_standin.py) -- in the build container only; the
       resulting logits are committed as fixtures under tests/golden/;
  (iii) gradients against torch autograd, AdamW against torch.optim.AdamW.
The reference ships no tests and no weights: end-to-end parity
against *trained* checkpoints is therefore UNPINNED; architecture-level parity is pinned
by (i)-(iii).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# timm model tables (convnext.py: convnext_pico / convnext_nano, conv_mlp=True)
ARCHS = {
    "convnext_pico": dict(depths=(2, 2, 6, 2), dims=(64, 128, 256, 512)),
    "convnext_nano": dict(depths=(2, 2, 8, 2), dims=(80, 160, 320, 640)),
}
LN_EPS = 1e-6
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def arch_of(model_kind: str) -> str:
    """'convnext_pico.d1_in1k', 'hf_hub:mwalmsley/zoobot-encoder-convnext_pico' -> table key."""
    mk = model_kind.lower()
    for name in ARCHS:
        if name in mk:
            return name
    raise ValueError(f"unsupported ConvNeXt model_kind: {model_kind}")


# --------------------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------------------
def layer_norm_c(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """LayerNorm2d: normalise over the channel dim of an NCHW tensor (eps 1e-6)."""
    y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, LN_EPS)
    return y.permute(0, 3, 1, 2)


def block(x: Tensor, sd: SD, p: str) -> Tensor:
    c = x.shape[1]
    y = F.conv2d(x, sd[p + "conv_dw.weight"], sd[p + "conv_dw.bias"], padding=3, groups=c)
    y = layer_norm_c(y, sd[p + "norm.weight"], sd[p + "norm.bias"])
    y = F.conv2d(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    y = F.gelu(y)  # exact erf form (nn.GELU() default)
    y = F.conv2d(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y * sd[p + "gamma"].reshape(1, -1, 1, 1)


def forward_features(img: Tensor, sd: SD, prefix: str, arch: str,
                     taps: Optional[dict] = None) -> Tensor:
    """img [B,3,H,W] -> [B,C3,h,w]; sd keys are ``prefix + 'stem.0.weight'`` etc."""
    depths = ARCHS[arch]["depths"]
    x = F.conv2d(img, sd[prefix + "stem.0.weight"], sd[prefix + "stem.0.bias"], stride=4)
    x = layer_norm_c(x, sd[prefix + "stem.1.weight"], sd[prefix + "stem.1.bias"])
    if taps is not None:
        taps["stem"] = x
    for i, depth in enumerate(depths):
        sp = f"{prefix}stages.{i}."
        if i > 0:
            x = layer_norm_c(x, sd[sp + "downsample.0.weight"], sd[sp + "downsample.0.bias"])
            x = F.conv2d(x, sd[sp + "downsample.1.weight"], sd[sp + "downsample.1.bias"], stride=2)
            if taps is not None:
                taps[f"down{i}"] = x
        for j in range(depth):
            x = block(x, sd, f"{sp}blocks.{j}.")
        if taps is not None:
            taps[f"stage{i}"] = x
    return x


def pooled_head(x: Tensor, w: Optional[Tensor], b: Optional[Tensor]) -> Tensor:
    """global avg pool -> (LayerNorm2d) -> flatten; architectures.py:109-112,137-141."""
    x = x.mean(dim=(2, 3), keepdim=True)
    if w is not None:
        x = layer_norm_c(x, w, b)
    return x.flatten(1)


# --------------------------------------------------------------------------------------
# heads
# --------------------------------------------------------------------------------------
def batch_norm1d(x: Tensor, sd: SD, p: str, training: bool, update: bool = True) -> Tensor:
    """nn.BatchNorm1d semantics: eval -> running stats; train -> batch stats (biased var for the
    normalisation, unbiased var into running_var, momentum 0.1)."""
    rm, rv = sd[p + "running_mean"], sd[p + "running_var"]
    if not training:
        return (x - rm) / torch.sqrt(rv + BN_EPS) * sd[p + "weight"] + sd[p + "bias"]
    mean = x.mean(0)
    var = x.var(0, unbiased=False)
    if update:
        n = x.shape[0]
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * x.detach().var(0, unbiased=True) if n > 1
                                          else BN_MOMENTUM * var.detach())
            if p + "num_batches_tracked" in sd:
                sd[p + "num_batches_tracked"] += 1
    return (x - mean) / torch.sqrt(var + BN_EPS) * sd[p + "weight"] + sd[p + "bias"]


def _act(kind: str):
    return F.gelu if kind == "gelu" else F.relu


def _drop(x: Tensor, p: float, mask: Optional[Tensor]) -> Tensor:
    """Inverted dropout with an explicit keep-mask (so the HIP path can be compared
    element-for-element); mask None -> identity (eval, or p == 0)."""
    if mask is None or p == 0.0:
        return x
    return x * mask / (1.0 - p)


def metadata_branch(meta: Tensor, sd: SD, p: str, act: str, trailing_act: bool,
                    training: bool = False, drop_p: float = 0.0,
                    drop_mask: Optional[Tensor] = None) -> Tensor:
    """BN1d -> Linear -> act -> Dropout -> Linear [-> act].
    mm_ConvNeXt: architectures.py:146-153 (GELU, trailing act, keys metadata_branch.{0,1,4});
    frozen_fusion's stripped um_nn: :282-290 + :299-303 (ReLU, NO trailing act,
    keys meta_branch.network.{0,1,4})."""
    a = _act(act)
    x = batch_norm1d(meta, sd, p + "0.", training)
    x = a(F.linear(x, sd[p + "1.weight"], sd[p + "1.bias"]))
    x = _drop(x, drop_p, drop_mask if training else None)
    x = F.linear(x, sd[p + "4.weight"], sd[p + "4.bias"])
    return a(x) if trailing_act else x


def fusion_head(feat: Tensor, sd: SD, p: str, act: str, training: bool = False,
                drop_p: float = 0.0, drop_mask: Optional[Tensor] = None) -> Tensor:
    """Linear -> act -> Linear -> act -> Dropout -> Linear(->1); keys {0,2,5}
    (architectures.py:156-164 GELU; :358-365 ReLU)."""
    a = _act(act)
    x = a(F.linear(feat, sd[p + "0.weight"], sd[p + "0.bias"]))
    x = a(F.linear(x, sd[p + "2.weight"], sd[p + "2.bias"]))
    x = _drop(x, drop_p, drop_mask if training else None)
    return F.linear(x, sd[p + "5.weight"], sd[p + "5.bias"])


# --------------------------------------------------------------------------------------
# whole models (same call shapes as the reference's nn.Modules)
# --------------------------------------------------------------------------------------
def mm_convnext_forward(sd: SD, config: dict, image: Tensor, meta: Tensor,
                        training: bool = False, masks: Optional[dict] = None,
                        taps: Optional[dict] = None) -> Tensor:
    """architectures.py:125-171.  Non-LS data: head is ``flatten`` only (needs a 1x1 map,
    true for 63x63 inputs); LS data: pool + head LayerNorm2d + flatten."""
    arch = arch_of(config.get("model_kind", "convnext_nano.d1h_in1k"))
    masks = masks or {}
    bp = "convnext_backbone."
    x = forward_features(image, sd, bp, arch, taps)
    if "LS" in config["train_data_version"]:
        f = pooled_head(x, sd[bp + "head.1.weight"], sd[bp + "head.1.bias"])
    else:
        f = x.flatten(1)
    m = metadata_branch(meta, sd, "metadata_branch.", "gelu", True, training,
                        config["meta_dropout"], masks.get("meta"))
    if taps is not None:
        taps["image_features"], taps["meta_features"] = f, m
    return fusion_head(torch.cat((f, m), dim=1), sd, "combined_head.", "gelu", training,
                       config["comb_dropout"], masks.get("comb"))


def convnext_forward(sd: SD, config: dict, image: Tensor, training: bool = False,
                     masks: Optional[dict] = None, prefix: str = "") -> Tensor:
    """architectures.py:104-122: pool, LayerNorm2d, flatten, Linear-GELU-Linear-GELU-Dropout-Linear
    (keys convnext.head.{1,3,5,8})."""
    arch = arch_of(config.get("model_kind", "convnext_nano.d1h_in1k"))
    masks = masks or {}
    bp = prefix + "convnext."
    x = forward_features(image, sd, bp, arch)
    f = pooled_head(x, sd[bp + "head.1.weight"], sd[bp + "head.1.bias"])
    x = F.gelu(F.linear(f, sd[bp + "head.3.weight"], sd[bp + "head.3.bias"]))
    x = F.gelu(F.linear(x, sd[bp + "head.5.weight"], sd[bp + "head.5.bias"]))
    x = _drop(x, config["dropout"], masks.get("head") if training else None)
    return F.linear(x, sd[bp + "head.8.weight"], sd[bp + "head.8.bias"])


def um_nn_forward(sd: SD, config: dict, meta: Tensor, training: bool = False,
                  masks: Optional[dict] = None, prefix: str = "") -> Tensor:
    """architectures.py:277-293 (keys network.{0,1,4,6})."""
    masks = masks or {}
    p = prefix + "network."
    x = metadata_branch(meta, sd, p, "relu", True, training, config["meta_dropout"],
                        masks.get("meta"))
    return F.linear(x, sd[p + "6.weight"], sd[p + "6.bias"])


def frozen_fusion_forward(sd: SD, config: dict, image: Tensor, meta: Tensor,
                          training: bool = False, masks: Optional[dict] = None) -> Tensor:
    """architectures.py:296-372 with a ConvNeXt image branch and a um_nn metadata branch:
    image_branch.convnext.* (head stripped to pool, LayerNorm2d, flatten  :309-313),
    meta_branch.network.{0,1,4} (:299-303), ReLU combined_head (:358-365)."""
    icfg = config["image_model_config"]
    mcfg = config["meta_model_config"]
    arch = arch_of(icfg.get("model_kind", "convnext_nano.d1h_in1k"))
    masks = masks or {}
    bp = "image_branch.convnext."
    x = forward_features(image, sd, bp, arch)
    f = pooled_head(x, sd[bp + "head.1.weight"], sd[bp + "head.1.bias"])
    m = metadata_branch(meta, sd, "meta_branch.network.", "relu", False, training,
                        mcfg["meta_dropout"], masks.get("meta"))
    return fusion_head(torch.cat((f, m), dim=1), sd, "combined_head.", "relu", training,
                       config["comb_dropout"], masks.get("comb"))


# --------------------------------------------------------------------------------------
# loss, optimiser, schedule
# --------------------------------------------------------------------------------------
def bce_with_logits(logits: Tensor, labels: Tensor, pos_weight: float) -> Tensor:
    """mean(-[w*y*log(sigmoid(z)) + (1-y)*log(1-sigmoid(z))]); train.py:211-212,525."""
    z, y = logits, labels
    return (-(pos_weight * y * F.logsigmoid(z) + (1 - y) * F.logsigmoid(-z))).mean()


def bce_grad(logits: Tensor, labels: Tensor, pos_weight: float, n_global: Optional[int] = None):
    """d(mean loss)/d(logits) in closed form: ((1-y) + w*y)*sigmoid(z) - w*y, over n."""
    n = n_global or logits.numel()
    s = torch.sigmoid(logits)
    return (((1 - labels) + pos_weight * labels) * s - pos_weight * labels) / n


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1: float, beta2: float, eps: float = 1e-8, wd: float = 1e-2) -> None:
    """torch.optim.AdamW (amsgrad False), in place; `step` is 1-based (train.py:242-246,527)."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def lr_sequence(lr: float, epochs: int, warmup_epochs: int) -> Sequence[float]:
    """LR used in epoch 0..epochs-1 by SequentialLR[LinearLR(0.01, total_iters=warmup),
    CosineAnnealingLR(T_max=max(1, epochs-warmup), eta_min=0.01*lr)] stepped per epoch
    (train.py:249-260,332).  Computed by running torch's own schedulers on a dummy
    optimiser so the torch-version quirks (warmup 0) are reproduced."""
    import warnings
    prm = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([prm], lr=lr)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sched = torch.optim.lr_scheduler.SequentialLR(
            opt,
            schedulers=[
                torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, total_iters=warmup_epochs),
                torch.optim.lr_scheduler.CosineAnnealingLR(
                    opt, T_max=max(1, epochs - warmup_epochs), eta_min=lr * 0.01),
            ],
            milestones=[warmup_epochs],
        )
        out = []
        for _ in range(epochs):
            out.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
    return out


# --------------------------------------------------------------------------------------
# seeded random weights (there are no bundled checkpoints and no network)
# --------------------------------------------------------------------------------------
def backbone_param_shapes(arch: str, prefix: str, head_norm: bool, head_norm_key: str = "head.1."):
    depths, dims = ARCHS[arch]["depths"], ARCHS[arch]["dims"]
    s = {prefix + "stem.0.weight": (dims[0], 3, 4, 4), prefix + "stem.0.bias": (dims[0],),
         prefix + "stem.1.weight": (dims[0],), prefix + "stem.1.bias": (dims[0],)}
    for i, (d, c) in enumerate(zip(depths, dims)):
        sp = f"{prefix}stages.{i}."
        if i > 0:
            cin = dims[i - 1]
            s[sp + "downsample.0.weight"] = (cin,)
            s[sp + "downsample.0.bias"] = (cin,)
            s[sp + "downsample.1.weight"] = (c, cin, 2, 2)
            s[sp + "downsample.1.bias"] = (c,)
        for j in range(d):
            bp = f"{sp}blocks.{j}."
            s[bp + "gamma"] = (c,)
            s[bp + "conv_dw.weight"] = (c, 1, 7, 7)
            s[bp + "conv_dw.bias"] = (c,)
            s[bp + "norm.weight"] = (c,)
            s[bp + "norm.bias"] = (c,)
            s[bp + "mlp.fc1.weight"] = (4 * c, c, 1, 1)
            s[bp + "mlp.fc1.bias"] = (4 * c,)
            s[bp + "mlp.fc2.weight"] = (c, 4 * c, 1, 1)
            s[bp + "mlp.fc2.bias"] = (c,)
    if head_norm:
        s[prefix + head_norm_key + "weight"] = (dims[-1],)
        s[prefix + head_norm_key + "bias"] = (dims[-1],)
    return s


def random_state_dict(shapes: dict, seed: int, gamma_value: float = 1.0) -> SD:
    """Deterministic 'trained-like' weights: fan-in scaled normal weights, small random biases,
    norm scales near 1, layer-scale gamma ~ gamma_value (NOT timm's 1e-6 init, so that every
    block contributes numerically; stated wherever numbers are reported), BN running stats
    non-trivial.  Key order is the dict order, so the stream is reproducible."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7, dtype=torch.long)
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(shp, generator=g) * 0.5
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith("gamma"):
            sd[k] = gamma_value * (1.0 + 0.1 * torch.randn(shp, generator=g))
        elif len(shp) == 1 and k.endswith("weight"):      # norm scales
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:                                # biases
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(fan_in)
    return sd


def head_param_shapes(kind: str, feat_dim: int, config: dict) -> dict:
    """Shapes of the non-backbone parameters for each wiring."""
    s = {}
    if kind in ("mm_ConvNeXt", "frozen_fusion"):
        nm = len(config["metadata_cols"]) if kind == "mm_ConvNeXt" else \
            len(config["meta_model_config"]["metadata_cols"])
        mc = config if kind == "mm_ConvNeXt" else config["meta_model_config"]
        mp = "metadata_branch." if kind == "mm_ConvNeXt" else "meta_branch.network."
        f1, f2 = mc["meta_fc1_neurons"], mc["meta_fc2_neurons"]
        s.update(_bn_shapes(mp + "0.", nm))
        s[mp + "1.weight"], s[mp + "1.bias"] = (f1, nm), (f1,)
        s[mp + "4.weight"], s[mp + "4.bias"] = (f2, f1), (f2,)
        c1, c2 = config["comb_fc1_neurons"], config["comb_fc2_neurons"]
        s["combined_head.0.weight"], s["combined_head.0.bias"] = (c1, feat_dim + f2), (c1,)
        s["combined_head.2.weight"], s["combined_head.2.bias"] = (c2, c1), (c2,)
        s["combined_head.5.weight"], s["combined_head.5.bias"] = (1, c2), (1,)
    elif kind == "ConvNeXt":
        f1, f2 = config["fc1_neurons"], config["fc2_neurons"]
        s["convnext.head.3.weight"], s["convnext.head.3.bias"] = (f1, feat_dim), (f1,)
        s["convnext.head.5.weight"], s["convnext.head.5.bias"] = (f2, f1), (f2,)
        s["convnext.head.8.weight"], s["convnext.head.8.bias"] = (1, f2), (1,)
    elif kind == "um_nn":
        nm = len(config["metadata_cols"])
        f1, f2 = config["meta_fc1_neurons"], config["meta_fc2_neurons"]
        s.update(_bn_shapes("network.0.", nm))
        s["network.1.weight"], s["network.1.bias"] = (f1, nm), (f1,)
        s["network.4.weight"], s["network.4.bias"] = (f2, f1), (f2,)
        s["network.6.weight"], s["network.6.bias"] = (1, f2), (1,)
    else:
        raise ValueError(kind)
    return s


def _bn_shapes(p: str, n: int) -> dict:
    return {p + "weight": (n,), p + "bias": (n,), p + "running_mean": (n,),
            p + "running_var": (n,), p + "num_batches_tracked": ()}


def model_param_shapes(kind: str, config: dict) -> dict:
    """Full state-dict shape table, in the reference's key order (backbone, heads)."""
    if kind == "mm_ConvNeXt":
        arch = arch_of(config.get("model_kind", "convnext_nano.d1h_in1k"))
        ls = "LS" in config["train_data_version"]
        s = backbone_param_shapes(arch, "convnext_backbone.", ls)
        s.update(head_param_shapes(kind, ARCHS[arch]["dims"][-1], config))
    elif kind == "ConvNeXt":
        arch = arch_of(config.get("model_kind", "convnext_nano.d1h_in1k"))
        s = backbone_param_shapes(arch, "convnext.", True)
        s.update(head_param_shapes(kind, ARCHS[arch]["dims"][-1], config))
    elif kind == "um_nn":
        s = head_param_shapes(kind, 0, config)
    elif kind == "frozen_fusion":
        icfg = config["image_model_config"]
        arch = arch_of(icfg.get("model_kind", "convnext_nano.d1h_in1k"))
        s = backbone_param_shapes(arch, "image_branch.convnext.", True)
        s.update(head_param_shapes(kind, ARCHS[arch]["dims"][-1], config))
    else:
        raise ValueError(kind)
    return s


def forward(kind: str, sd: SD, config: dict, image: Optional[Tensor], meta: Optional[Tensor],
            training: bool = False, masks: Optional[dict] = None) -> Tensor:
    if kind == "mm_ConvNeXt":
        return mm_convnext_forward(sd, config, image, meta, training, masks)
    if kind == "ConvNeXt":
        return convnext_forward(sd, config, image, training, masks)
    if kind == "um_nn":
        return um_nn_forward(sd, config, meta, training, masks)
    if kind == "frozen_fusion":
        return frozen_fusion_forward(sd, config, image, meta, training, masks)
    raise ValueError(kind)
