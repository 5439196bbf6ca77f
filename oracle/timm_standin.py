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


def create_model(model_kind, pretrained=False, **kw):
    if pretrained:
        raise RuntimeError("stand-in timm has no pretrained weights (no network in this image)")
    for name, (depths, dims) in _TABLE.items():
        if name in model_kind.lower():
            return ConvNeXtStandIn(depths, dims)
    raise RuntimeError(f"stand-in timm: unsupported model {model_kind}")


def install():
    sys.modules["timm"] = sys.modules[__name__]
