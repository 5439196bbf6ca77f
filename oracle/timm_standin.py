"""Stand-in for the ``timm`` package so that the reference's own wrapper classes
(/root/reference/btsbot/architectures.py) can be imported and run in the build container.

TEST INFRASTRUCTURE.  timm is an un-vendored, unpinned (>=0.9.0) dependency of the reference
and is not installed in this image.  This module re-creates, as
``nn.Module``s, only the surface the reference touches:

    timm.create_model(model_kind, pretrained=...)             architectures.py:108,132
    backbone.head.in_features / .global_pool / .norm / .flatten   architectures.py:110-113,134-143
    backbone(x)  == head(forward_features(x))

with the parameter names timm 1.0 gives ConvNeXt (stem.0/1, stages.i.downsample.0/1,
stages.i.blocks.j.{gamma,conv_dw,norm,mlp.fc1,mlp.fc2}, head.norm, head.fc).  The maths is the
published ConvNeXt-v1 block; it is written independently of oracle/convnext_oracle.py
(module form vs functional form) so the two cross-check each other, and both are checked
against ``transformers.ConvNextModel``.

``MaxxVitStandIn`` does the same for ``maxvit_tiny_rw_224`` (architectures.py:31,62: ``.head.in_features``,
``.head.global_pool``, ``backbone(x)``), with timm's MaxxVit parameter names; see
oracle/maxvit_oracle.py for what is and is not pinned about that definition.

Use:  ``install()`` puts this module in ``sys.modules['timm']``.
"""
from __future__ import annotations

import sys

import torch
import torch.nn as nn

_TABLE = {
    "convnext_pico": ((2, 2, 6, 2), (64, 128, 256, 512)),
    "convnext_nano": ((2, 2, 8, 2), (80, 160, 320, 640)),
}


class LayerNorm2d(nn.LayerNorm):
    def __init__(self, c, eps=1e-6):
        super().__init__(c, eps=eps)

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        x = nn.functional.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class _ConvMlp(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc1 = nn.Conv2d(c, 4 * c, 1)
        self.act = nn.GELU()
        self.fc2 = nn.Conv2d(4 * c, c, 1)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, c, ls_init=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(c, c, 7, padding=3, groups=c)
        self.norm = LayerNorm2d(c)
        self.mlp = _ConvMlp(c)
        self.gamma = nn.Parameter(ls_init * torch.ones(c))

    def forward(self, x):
        return x + self.mlp(self.norm(self.conv_dw(x))) * self.gamma.reshape(1, -1, 1, 1)


class _Stage(nn.Module):
    def __init__(self, cin, cout, depth, first):
        super().__init__()
        if first:
            self.downsample = nn.Identity()
        else:
            self.downsample = nn.Sequential(LayerNorm2d(cin), nn.Conv2d(cin, cout, 2, stride=2))
        self.blocks = nn.Sequential(*[_Block(cout) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class _Head(nn.Module):
    def __init__(self, c, num_classes=1000):
        super().__init__()
        self.in_features = c
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.norm = LayerNorm2d(c)
        self.flatten = nn.Flatten(1)
        self.drop = nn.Dropout(0.0)
        self.fc = nn.Linear(c, num_classes)

    def forward(self, x):
        return self.fc(self.drop(self.flatten(self.norm(self.global_pool(x)))))


class ConvNeXtStandIn(nn.Module):
    def __init__(self, depths, dims):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, dims[0], 4, stride=4), LayerNorm2d(dims[0]))
        stages, cin = [], dims[0]
        for i, (d, c) in enumerate(zip(depths, dims)):
            stages.append(_Stage(cin, c, d, first=(i == 0)))
            cin = c
        self.stages = nn.Sequential(*stages)
        self.norm_pre = nn.Identity()
        self.head = _Head(dims[-1])

    def forward_features(self, x):
        return self.norm_pre(self.stages(self.stem(x)))

    def forward(self, x):
        return self.head(self.forward_features(x))


# --------------------------------------------------------------------------------------
# MaxViT ("maxvit_tiny_rw_224"): module form with timm's MaxxVit parameter names
# (stem.conv1/norm1/conv2, stages.i.blocks.j.{conv,attn_block,attn_grid}.*, norm, head.fc), written
# with nn.Modules + einsum attention, independently of the functional oracle/maxvit_oracle.py.
# --------------------------------------------------------------------------------------
class _BnAct(nn.BatchNorm2d):
    """timm BatchNormAct2d: BatchNorm2d parameters, optional SiLU."""

    def __init__(self, c, act=True):
        super().__init__(c, eps=1e-5)
        self.apply_act = act

    def forward(self, x):
        x = super().forward(x)
        return nn.functional.silu(x) if self.apply_act else x


class _SE(nn.Module):
    def __init__(self, c, rd):
        super().__init__()
        self.fc1 = nn.Conv2d(c, rd, 1)
        self.fc2 = nn.Conv2d(rd, c, 1)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * torch.sigmoid(self.fc2(nn.functional.silu(self.fc1(s))))


class _Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.pool = nn.AvgPool2d(2)
        self.expand = nn.Conv2d(cin, cout, 1, bias=False) if cin != cout else nn.Identity()

    def forward(self, x):
        return self.expand(self.pool(x))


class _MbConv(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        mid = 4 * cin
        self.shortcut = _Down(cin, cout) if stride == 2 else nn.Identity()
        self.pre_norm = _BnAct(cin, act=False)
        self.down = nn.Identity()
        self.conv1_1x1 = nn.Conv2d(cin, mid, 1)
        self.norm1 = _BnAct(mid)
        self.conv2_kxk = nn.Conv2d(mid, mid, 3, stride=stride, padding=1, groups=mid)
        self.norm2 = _BnAct(mid)
        self.se = _SE(mid, mid // 16)
        self.conv3_1x1 = nn.Conv2d(mid, cout, 1, bias=False)

    def forward(self, x):
        y = self.norm1(self.conv1_1x1(self.pre_norm(x)))
        y = self.se(self.norm2(self.conv2_kxk(y)))
        return self.conv3_1x1(y) + self.shortcut(x)


class _RelPosBias(nn.Module):
    def __init__(self, ws, heads):
        super().__init__()
        self.ws, self.heads = ws, heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        idx = torch.empty(ws * ws, ws * ws, dtype=torch.long)
        for i in range(ws * ws):
            for j in range(ws * ws):
                dy, dx = i // ws - j // ws, i % ws - j % ws
                idx[i, j] = (dy + ws - 1) * (2 * ws - 1) + dx + ws - 1
        self.register_buffer("relative_position_index", idx.view(-1), persistent=False)

    def get_bias(self):
        n = self.ws * self.ws
        return self.relative_position_bias_table[self.relative_position_index].view(n, n, -1) \
            .permute(2, 0, 1)


class _Attn(nn.Module):
    def __init__(self, c, ws):
        super().__init__()
        self.heads = c // 32
        self.qkv = nn.Linear(c, 3 * c)
        self.rel_pos = _RelPosBias(ws, self.heads)
        self.proj = nn.Linear(c, c)

    def forward(self, x):                      # [nW, N, C]
        nw, n, c = x.shape
        qkv = self.qkv(x).reshape(nw, n, self.heads, 3, 32)      # head-first channel order
        q, k, v = qkv[:, :, :, 0], qkv[:, :, :, 1], qkv[:, :, :, 2]
        a = torch.einsum("wihd,wjhd->whij", q * 32 ** -0.5, k) + self.rel_pos.get_bias()[None]
        o = torch.einsum("whij,wjhd->wihd", a.softmax(-1), v).reshape(nw, n, c)
        return self.proj(o)


class _Mlp(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc1 = nn.Linear(c, 4 * c)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(4 * c, c)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _PartitionAttn(nn.Module):
    def __init__(self, c, grid, ws=7):
        super().__init__()
        self.grid, self.ws = grid, ws
        self.norm1 = nn.LayerNorm(c, eps=1e-6)
        self.attn = _Attn(c, ws)
        self.norm2 = nn.LayerNorm(c, eps=1e-6)
        self.mlp = _Mlp(c)

    def forward(self, x):                      # NHWC
        b, h, w, c = x.shape
        ws = self.ws
        y = self.norm1(x)
        if self.grid:      # token (gy, gx) of window (iy, ix) = pixel (gy * h/ws + iy, gx * w/ws + ix)
            y = y.reshape(b, ws, h // ws, ws, w // ws, c).permute(0, 2, 4, 1, 3, 5)
        else:
            y = y.reshape(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5)
        y = self.attn(y.reshape(-1, ws * ws, c)).reshape(b, h // ws, w // ws, ws, ws, c)
        if self.grid:
            y = y.permute(0, 3, 1, 4, 2, 5)
        else:
            y = y.permute(0, 1, 3, 2, 4, 5)
        x = x + y.reshape(b, h, w, c)
        return x + self.mlp(self.norm2(x))


class _MaxBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = _MbConv(cin, cout, stride)
        self.attn_block = _PartitionAttn(cout, grid=False)
        self.attn_grid = _PartitionAttn(cout, grid=True)

    def forward(self, x):
        x = self.conv(x).permute(0, 2, 3, 1)
        return self.attn_grid(self.attn_block(x)).permute(0, 3, 1, 2)


class _MaxStage(nn.Module):
    def __init__(self, cin, cout, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[_MaxBlock(cin if j == 0 else cout, cout, 2 if j == 0 else 1)
                                      for j in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class _MaxStem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 32, 3, stride=2, padding=1, bias=False)
        self.norm1 = _BnAct(32)
        self.conv2 = nn.Conv2d(32, 64, 3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return self.conv2(self.norm1(self.conv1(x)))


class _FlatPool(nn.Module):
    """timm SelectAdaptivePool2d(pool_type='avg', flatten=True)."""

    def forward(self, x):
        return x.mean((2, 3))


class _MaxHead(nn.Module):
    def __init__(self, c, num_classes=1000):
        super().__init__()
        self.in_features = c
        self.global_pool = _FlatPool()
        self.drop = nn.Dropout(0.0)
        self.fc = nn.Linear(c, num_classes)

    def forward(self, x):
        return self.fc(self.drop(self.global_pool(x)))


class MaxxVitStandIn(nn.Module):
    def __init__(self, depths=(2, 2, 5, 2), dims=(64, 128, 256, 512)):
        super().__init__()
        self.stem = _MaxStem()
        stages, cin = [], 64
        for d, c in zip(depths, dims):
            stages.append(_MaxStage(cin, c, d))
            cin = c
        self.stages = nn.Sequential(*stages)
        self.norm = LayerNorm2d(dims[-1], eps=1e-6)
        self.head = _MaxHead(dims[-1])

    def forward_features(self, x):
        return self.norm(self.stages(self.stem(x)))

    def forward(self, x):
        return self.head(self.forward_features(x))


def create_model(model_kind, pretrained=False, **kw):
    if pretrained:
        raise RuntimeError("stand-in timm has no pretrained weights (no network in this image)")
    for name, (depths, dims) in _TABLE.items():
        if name in model_kind.lower():
            return ConvNeXtStandIn(depths, dims)
    if "maxvit_tiny_rw_224" in model_kind.lower():
        return MaxxVitStandIn()
    raise RuntimeError(f"stand-in timm: unsupported model {model_kind}")


def install():
    sys.modules["timm"] = sys.modules[__name__]
