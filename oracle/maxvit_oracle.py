"""fp32 CPU restatement of BTSbot's MaxViT image branch.

TEST INFRASTRUCTURE (checker + CPU baseline) -- never imported by the product package.

What it restates (paths relative to /root/reference):
  * ``timm.create_model("maxvit_tiny_rw_224.sw_in1k")`` reached at btsbot/architectures.py:31,62.
    timm (>=0.9.0, floor only, pyproject.toml:43) is NOT in the reference tree and not installed
    here, so the algorithm below is the published MaxViT block (Tu et al. 2022) in timm's "rw"
    variant, written from the model definition as timm 0.9/1.0 publishes it
    (timm/models/maxxvit.py: MaxxVit, Stem, MaxxVitStage, MaxxVitBlock, MbConvBlock,
    PartitionAttentionCl, AttentionCl; timm/layers: SEModule, RelPosBias, Downsample2d):
      stem    conv3x3 s2 (3->32, no bias) . BN . SiLU . conv3x3 s1 (32->64, no bias)      224 -> 112
      stage i (dims 64,128,256,512; depths 2,2,5,2; first block of each stage stride 2), block =
        MBConv   sc + conv3_1x1( SE( SiLU(BN( dw3x3_s( SiLU(BN( conv1_1x1( BN(x) )))))))) )
                 mid = 4 * C_in ("rw": expansion from the INPUT width), stride in the depthwise
                 conv, SE reduction = mid/16 with SiLU, conv3 without bias;
                 sc = x (stride 1) or conv1x1_nobias(avgpool2x2(x)) (stride 2; no conv if C_in == C_out)
        window attention  x + proj(MHSA_7x7windows(LN(x)));  x + fc2(GELU(fc1(LN(x))))
        grid attention    the same on the 7x7 dilated grid partition
        MHSA: dim_head 32, heads = C/32, qkv channel order head-first ([head][q|k|v][32]),
              logits = (q*32^-0.5) k^T + B[h], B from a learned (13*13, heads) table indexed by
              relative offset (Swin indexing), softmax, proj.  LayerNorm eps 1e-6, BatchNorm eps 1e-5.
      final   LayerNorm2d(512) -> global average pool (the reference keeps head.global_pool only,
              architectures.py:65) -> [B,512]
  * the wrappers btsbot/architectures.py:25-55 (MaxViT) and :58-101 (mm_MaxViT): bilinear resize to
    224 with align_corners=False (:44-50,:90-96), heads.

Cross-checks available without timm (tests/test_oracle_pins.py): the parameter count of this
definition is 28,562,232 without the 1000-class fc = 29,075,232 with it (timm's model table lists
maxvit_tiny_rw_224 at 29.1 M) and its multiply-accumulate count at 224x224 is 5.07 G (table: 5.1 G).
Beyond that **parity is UNPINNED for this row**: no timm, no weights, no reference tests; biases of
conv1_1x1 / conv2_kxk, the SE activation and the head-first qkv order are recalled, not verified
(SURVEY.md 8 a7 calls this the lowest-confidence row).  The wrapper wiring IS pinned: the
reference's own MaxViT / mm_MaxViT classes run around the module-form stand-in
(oracle/timm_standin.py) in tests/golden/make_golden.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import convnext_oracle as CO

Tensor = torch.Tensor
SD = Dict[str, Tensor]

ARCHS = {
    "maxvit_tiny_rw_224": dict(depths=(2, 2, 5, 2), dims=(64, 128, 256, 512), stem=(32, 64),
                               img=224, window=7, dim_head=32, se_ratio=1.0 / 16, expand=4),
}
LN_EPS = 1e-6
BN_EPS = 1e-5


def arch_of(model_kind: str) -> str:
    mk = model_kind.lower()
    for name in ARCHS:
        if name in mk:
            return name
    raise ValueError(f"unsupported MaxViT model_kind: {model_kind}")


def block_table(arch: str):
    """[(stage, block, c_in, c_out, mid, rd, stride, hw_in, hw_out)] for every MaxxVitBlock."""
    a = ARCHS[arch]
    rows = []
    cin, hw = a["stem"][1], a["img"] // 2
    for i, (d, c) in enumerate(zip(a["depths"], a["dims"])):
        for j in range(d):
            stride = 2 if j == 0 else 1
            mid = a["expand"] * cin
            rows.append((i, j, cin, c, mid, int(a["se_ratio"] * mid), stride, hw, hw // stride))
            cin, hw = c, hw // stride
    return rows


# --------------------------------------------------------------------------------------
# pieces
# --------------------------------------------------------------------------------------
def bn2d(x: Tensor, sd: SD, p: str, batch_stats: bool = False, new_stats: Optional[dict] = None) -> Tensor:
    """BatchNorm2d: eval mode (running statistics), or -- batch_stats, the training mode the reference fine-tunes the
    branch in (train.py:218-236: every parameter trainable, model.train()) -- batch statistics, with the running
    statistics torch would leave behind (momentum 0.1, unbiased variance) collected in `new_stats`."""
    if not batch_stats:
        return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"],
                            sd[p + "bias"], False, 0.0, BN_EPS)
    rm, rv = sd[p + "running_mean"].detach().clone(), sd[p + "running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], True, 0.1, BN_EPS)
    if new_stats is not None:
        new_stats[p + "running_mean"], new_stats[p + "running_var"] = rm, rv
    return y


def rel_pos_index(ws: int) -> Tensor:
    """Swin-style relative position index of a ws x ws window: [(ws*ws), (ws*ws)] int64,
    idx[i][j] = (yi - yj + ws-1) * (2ws-1) + (xi - xj + ws-1)."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def window_partition(x: Tensor, ws: int) -> Tensor:
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)


def window_reverse(win: Tensor, ws: int, h: int, w: int) -> Tensor:
    c = win.shape[-1]
    x = win.view(-1, h // ws, w // ws, ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, h, w, c)


def grid_partition(x: Tensor, gs: int) -> Tensor:
    b, h, w, c = x.shape
    x = x.view(b, gs, h // gs, gs, w // gs, c)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(-1, gs, gs, c)


def grid_reverse(win: Tensor, gs: int, h: int, w: int) -> Tensor:
    c = win.shape[-1]
    x = win.view(-1, h // gs, w // gs, gs, gs, c)
    return x.permute(0, 3, 1, 4, 2, 5).reshape(-1, h, w, c)


def attention_cl(x: Tensor, sd: SD, p: str, dim_head: int, ws: int) -> Tensor:
    """x [nW, ws, ws, C] -> same; head-first qkv, learned relative position bias."""
    nw = x.shape[0]
    c = x.shape[-1]
    heads = c // dim_head
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    q, k, v = qkv.view(nw, -1, heads, dim_head * 3).transpose(1, 2).chunk(3, dim=3)
    bias = sd[p + "rel_pos.relative_position_bias_table"][rel_pos_index(ws).view(-1)]
    bias = bias.view(ws * ws, ws * ws, heads).permute(2, 0, 1).unsqueeze(0)
    attn = (q * dim_head ** -0.5) @ k.transpose(-2, -1) + bias
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(nw, ws, ws, c)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def partition_attention(x: Tensor, sd: SD, p: str, grid: bool, dim_head: int, ws: int) -> Tensor:
    """PartitionAttentionCl on an NHWC map."""
    _b, h, w, c = x.shape
    y = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
    part = grid_partition(y, ws) if grid else window_partition(y, ws)
    part = attention_cl(part, sd, p + "attn.", dim_head, ws)
    y = grid_reverse(part, ws, h, w) if grid else window_reverse(part, ws, h, w)
    x = x + y
    y = F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
    y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def mbconv(x: Tensor, sd: SD, p: str, stride: int, batch_stats: bool = False,
           new_stats: Optional[dict] = None) -> Tensor:
    """MbConvBlock, NCHW."""
    if stride == 2:
        sc = F.avg_pool2d(x, 2)
        if p + "shortcut.expand.weight" in sd:
            sc = F.conv2d(sc, sd[p + "shortcut.expand.weight"])
    else:
        sc = x
    y = bn2d(x, sd, p + "pre_norm.", batch_stats, new_stats)
    y = F.conv2d(y, sd[p + "conv1_1x1.weight"], sd[p + "conv1_1x1.bias"])
    y = F.silu(bn2d(y, sd, p + "norm1.", batch_stats, new_stats))
    mid = y.shape[1]
    y = F.conv2d(y, sd[p + "conv2_kxk.weight"], sd[p + "conv2_kxk.bias"], stride=stride, padding=1,
                 groups=mid)
    y = F.silu(bn2d(y, sd, p + "norm2.", batch_stats, new_stats))
    s = y.mean((2, 3), keepdim=True)
    s = F.silu(F.conv2d(s, sd[p + "se.fc1.weight"], sd[p + "se.fc1.bias"]))
    s = torch.sigmoid(F.conv2d(s, sd[p + "se.fc2.weight"], sd[p + "se.fc2.bias"]))
    y = y * s
    y = F.conv2d(y, sd[p + "conv3_1x1.weight"])
    return y + sc


def resize(img: Tensor, size: int) -> Tensor:
    """architectures.py:44-50 / :90-96."""
    if img.shape[-1] != size or img.shape[-2] != size:
        img = F.interpolate(img, size=(size, size), mode="bilinear", align_corners=False)
    return img


def forward_features(img: Tensor, sd: SD, prefix: str, arch: str,
                     taps: Optional[dict] = None, batch_stats: bool = False,
                     new_stats: Optional[dict] = None) -> Tensor:
    """img [B,3,224,224] -> [B,512,7,7] (after the final LayerNorm2d).  batch_stats: BatchNorm2d in training mode."""
    a = ARCHS[arch]
    x = F.conv2d(img, sd[prefix + "stem.conv1.weight"], None, stride=2, padding=1)
    x = F.silu(bn2d(x, sd, prefix + "stem.norm1.", batch_stats, new_stats))
    x = F.conv2d(x, sd[prefix + "stem.conv2.weight"], None, stride=1, padding=1)
    if taps is not None:
        taps["stem"] = x
    for (i, j, _cin, _c, _mid, _rd, stride, _hi, _ho) in block_table(arch):
        p = f"{prefix}stages.{i}.blocks.{j}."
        x = mbconv(x, sd, p + "conv.", stride, batch_stats, new_stats)
        if taps is not None:
            taps[f"s{i}b{j}.conv"] = x
        x = x.permute(0, 2, 3, 1)
        x = partition_attention(x, sd, p + "attn_block.", False, a["dim_head"], a["window"])
        if taps is not None:
            taps[f"s{i}b{j}.block"] = x.permute(0, 3, 1, 2)
        x = partition_attention(x, sd, p + "attn_grid.", True, a["dim_head"], a["window"])
        x = x.permute(0, 3, 1, 2)
        if taps is not None:
            taps[f"s{i}b{j}"] = x
    c = x.shape[1]
    x = F.layer_norm(x.permute(0, 2, 3, 1), (c,), sd[prefix + "norm.weight"],
                     sd[prefix + "norm.bias"], LN_EPS).permute(0, 3, 1, 2)
    return x


def pooled(x: Tensor) -> Tensor:
    """head.global_pool = SelectAdaptivePool2d('avg', flatten=True)."""
    return x.mean((2, 3))


# --------------------------------------------------------------------------------------
# whole models
# --------------------------------------------------------------------------------------
def mm_maxvit_forward(sd: SD, config: dict, image: Tensor, meta: Tensor, training: bool = False,
                      masks: Optional[dict] = None, taps: Optional[dict] = None, branch_training: bool = False,
                      new_stats: Optional[dict] = None) -> Tensor:
    """architectures.py:58-101 (keys maxvit_backbone.*, metadata_branch.{0,1,4},
    combined_head.{0,2,5}).  The image branch has BatchNorm2d: eval mode (running statistics) unless
    branch_training, the mode the reference trains the whole model in (batch statistics)."""
    arch = arch_of(config.get("model_kind", "maxvit_tiny_rw_224.sw_in1k"))
    masks = masks or {}
    x = resize(image, ARCHS[arch]["img"])
    f = pooled(forward_features(x, sd, "maxvit_backbone.", arch, taps, branch_training, new_stats))
    m = CO.metadata_branch(meta, sd, "metadata_branch.", "gelu", True, training,
                           config["meta_dropout"], masks.get("meta"))
    if taps is not None:
        taps["image_features"], taps["meta_features"] = f, m
    return CO.fusion_head(torch.cat((f, m), dim=1), sd, "combined_head.", "gelu", training,
                          config["comb_dropout"], masks.get("comb"))


def maxvit_forward(sd: SD, config: dict, image: Tensor, training: bool = False,
                   masks: Optional[dict] = None, branch_training: bool = False,
                   new_stats: Optional[dict] = None) -> Tensor:
    """architectures.py:25-55: head = global_pool, Linear, GELU, Linear, GELU, Dropout, Linear
    (keys maxvit.head.{1,3,6})."""
    arch = arch_of(config.get("model_kind", "maxvit_tiny_rw_224.sw_in1k"))
    masks = masks or {}
    bp = "maxvit."
    x = resize(image, ARCHS[arch]["img"])
    f = pooled(forward_features(x, sd, bp, arch, None, branch_training, new_stats))
    x = F.gelu(F.linear(f, sd[bp + "head.1.weight"], sd[bp + "head.1.bias"]))
    x = F.gelu(F.linear(x, sd[bp + "head.3.weight"], sd[bp + "head.3.bias"]))
    x = CO._drop(x, config["dropout"], masks.get("head") if training else None)
    return F.linear(x, sd[bp + "head.6.weight"], sd[bp + "head.6.bias"])


def frozen_fusion_maxvit_forward(sd: SD, config: dict, image: Tensor, meta: Tensor,
                                 training: bool = False, masks: Optional[dict] = None) -> Tensor:
    """architectures.py:296-372 with a MaxViT image branch (head stripped to its global pool, :304-308)
    and a um_nn metadata branch (:299-303): keys image_branch.maxvit.*, meta_branch.network.{0,1,4},
    ReLU combined_head.{0,2,5}."""
    icfg, mcfg = config["image_model_config"], config["meta_model_config"]
    arch = arch_of(icfg.get("model_kind", "maxvit_tiny_rw_224.sw_in1k"))
    masks = masks or {}
    x = resize(image, ARCHS[arch]["img"])
    f = pooled(forward_features(x, sd, "image_branch.maxvit.", arch))
    m = CO.metadata_branch(meta, sd, "meta_branch.network.", "relu", False, training,
                           mcfg["meta_dropout"], masks.get("meta"))
    return CO.fusion_head(torch.cat((f, m), dim=1), sd, "combined_head.", "relu", training,
                          config["comb_dropout"], masks.get("comb"))


# --------------------------------------------------------------------------------------
# parameter tables
# --------------------------------------------------------------------------------------
def _bn(p: str, n: int) -> dict:
    return {p + "weight": (n,), p + "bias": (n,), p + "running_mean": (n,),
            p + "running_var": (n,), p + "num_batches_tracked": ()}


def _attn_shapes(p: str, c: int, dim_head: int, ws: int) -> dict:
    heads = c // dim_head
    return {
        p + "norm1.weight": (c,), p + "norm1.bias": (c,),
        p + "attn.qkv.weight": (3 * c, c), p + "attn.qkv.bias": (3 * c,),
        p + "attn.rel_pos.relative_position_bias_table": ((2 * ws - 1) ** 2, heads),
        p + "attn.proj.weight": (c, c), p + "attn.proj.bias": (c,),
        p + "norm2.weight": (c,), p + "norm2.bias": (c,),
        p + "mlp.fc1.weight": (4 * c, c), p + "mlp.fc1.bias": (4 * c,),
        p + "mlp.fc2.weight": (c, 4 * c), p + "mlp.fc2.bias": (c,),
    }


def backbone_param_shapes(arch: str, prefix: str) -> dict:
    """timm state-dict keys of MaxxVit (module registration order), without head.fc."""
    a = ARCHS[arch]
    s0, s1 = a["stem"]
    s = {prefix + "stem.conv1.weight": (s0, 3, 3, 3)}
    s.update(_bn(prefix + "stem.norm1.", s0))
    s[prefix + "stem.conv2.weight"] = (s1, s0, 3, 3)
    for (i, j, cin, c, mid, rd, stride, _hi, _ho) in block_table(arch):
        p = f"{prefix}stages.{i}.blocks.{j}.conv."
        if stride == 2 and cin != c:
            s[p + "shortcut.expand.weight"] = (c, cin, 1, 1)
        s.update(_bn(p + "pre_norm.", cin))
        s[p + "conv1_1x1.weight"], s[p + "conv1_1x1.bias"] = (mid, cin, 1, 1), (mid,)
        s.update(_bn(p + "norm1.", mid))
        s[p + "conv2_kxk.weight"], s[p + "conv2_kxk.bias"] = (mid, 1, 3, 3), (mid,)
        s.update(_bn(p + "norm2.", mid))
        s[p + "se.fc1.weight"], s[p + "se.fc1.bias"] = (rd, mid, 1, 1), (rd,)
        s[p + "se.fc2.weight"], s[p + "se.fc2.bias"] = (mid, rd, 1, 1), (mid,)
        s[p + "conv3_1x1.weight"] = (c, mid, 1, 1)
        bp = f"{prefix}stages.{i}.blocks.{j}."
        s.update(_attn_shapes(bp + "attn_block.", c, a["dim_head"], a["window"]))
        s.update(_attn_shapes(bp + "attn_grid.", c, a["dim_head"], a["window"]))
    s[prefix + "norm.weight"] = (a["dims"][-1],)
    s[prefix + "norm.bias"] = (a["dims"][-1],)
    return s


def model_param_shapes(kind: str, config: dict) -> dict:
    arch = arch_of(config.get("model_kind", "maxvit_tiny_rw_224.sw_in1k"))
    feat = ARCHS[arch]["dims"][-1]
    if kind == "mm_MaxViT":
        s = backbone_param_shapes(arch, "maxvit_backbone.")
        s.update(CO.head_param_shapes("mm_ConvNeXt", feat, config))
    elif kind == "MaxViT":
        s = backbone_param_shapes(arch, "maxvit.")
        f1, f2 = config["fc1_neurons"], config["fc2_neurons"]
        s["maxvit.head.1.weight"], s["maxvit.head.1.bias"] = (f1, feat), (f1,)
        s["maxvit.head.3.weight"], s["maxvit.head.3.bias"] = (f2, f1), (f2,)
        s["maxvit.head.6.weight"], s["maxvit.head.6.bias"] = (1, f2), (1,)
    elif kind == "frozen_fusion":
        icfg = config["image_model_config"]
        arch = arch_of(icfg.get("model_kind", "maxvit_tiny_rw_224.sw_in1k"))
        s = backbone_param_shapes(arch, "image_branch.maxvit.")
        s.update(CO.head_param_shapes("frozen_fusion", ARCHS[arch]["dims"][-1], config))
    else:
        raise ValueError(kind)
    return s


def count_params(arch: str) -> int:
    n = 0
    for k, shp in backbone_param_shapes(arch, "").items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        m = 1
        for d in shp:
            m *= d
        n += m
    return n


def count_macs(arch: str) -> int:
    """Multiply-accumulates of the backbone at the native input size (convs, linears, attention)."""
    a = ARCHS[arch]
    hw = a["img"] // 2
    macs = hw * hw * (27 * a["stem"][0] + 9 * a["stem"][0] * a["stem"][1])
    ws2 = a["window"] ** 2
    for (_i, _j, cin, c, mid, rd, stride, hi, ho) in block_table(arch):
        macs += hi * hi * cin * mid + ho * ho * 9 * mid + 2 * mid * rd + ho * ho * mid * c
        if stride == 2 and cin != c:
            macs += ho * ho * cin * c
        per_attn = ho * ho * (3 * c * c + c * c + 8 * c * c) + ho * ho * ws2 * c * 2
        macs += 2 * per_attn
    return macs


def forward(kind: str, sd: SD, config: dict, image: Optional[Tensor], meta: Optional[Tensor],
            training: bool = False, masks: Optional[dict] = None, branch_training: bool = False,
            new_stats: Optional[dict] = None) -> Tensor:
    if kind == "mm_MaxViT":
        return mm_maxvit_forward(sd, config, image, meta, training, masks, None, branch_training, new_stats)
    if kind == "MaxViT":
        return maxvit_forward(sd, config, image, training, masks, branch_training, new_stats)
    if branch_training:
        raise NotImplementedError("branch_training is restated for mm_MaxViT and MaxViT")
    if kind == "frozen_fusion":
        return frozen_fusion_maxvit_forward(sd, config, image, meta, training, masks)
    raise ValueError(kind)
