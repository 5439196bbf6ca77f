"""CPU emulation of the fp8 (OCP e4m3) operand mode of stages 2-3 under different scaling schemes -- what the block scales of
v_mfma_scale_f32_*_f8f6f4 (lane map: tools/unit/mx_probe2.hip) could buy before any kernel is written.
Only the fp8 roundings of the pointwise convolutions of stages 2-3 are emulated (everything else fp32), so the figures are
the share of the mode's error that scaling can move.
  per_filter : one power-of-two scale per filter row, activations clamped to +-448 unscaled   (the shipped mode)
  block_w    : + one E8M0 scale per (filter row, 32 input channels)
  block_wx   : + one E8M0 scale per (pixel, 32 input channels) of the LayerNorm output / hidden activation
usage: python tools/fp8_emul.py [alerts]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn.functional as F
from helpers import CONFIGS, seeded_state
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O

E4M3_MAX = 448.0


def q8(x):
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float()


def pow2_scale(amax):   # scale s = 2^k with amax * s in (224, 448]
    amax = amax.clamp_min(1e-30)
    return torch.exp2(torch.floor(torch.log2(E4M3_MAX / amax)))


def quant_rows(w, block):   # w [N, K]; block None: per-row scale; else per (row, block of K)
    if block is None:
        s = pow2_scale(w.abs().amax(1, keepdim=True))
        return q8(w * s) / s
    n, k = w.shape
    wb = w.reshape(n, k // block, block)
    s = pow2_scale(wb.abs().amax(2, keepdim=True))
    return (q8(wb * s) / s).reshape(n, k)


def qconv(x, w, b, scheme):   # x [B,C,H,W], w [N,C,1,1]
    n, c = w.shape[:2]
    w2 = w.reshape(n, c)
    xr = x.permute(0, 2, 3, 1).reshape(-1, c)
    if scheme == "per_filter":
        wq, xq = quant_rows(w2, None), q8(xr)
    elif scheme == "block_w":
        wq, xq = quant_rows(w2, 32), q8(xr)
    else:
        wq, xq = quant_rows(w2, 32), quant_rows(xr, 32)
    y = xq @ wq.t() + b
    return y.reshape(x.shape[0], x.shape[2], x.shape[3], n).permute(0, 3, 1, 2)


def block(x, sd, p, scheme):
    c = x.shape[1]
    y = F.conv2d(x, sd[p + "conv_dw.weight"], sd[p + "conv_dw.bias"], padding=3, groups=c)
    y = O.layer_norm_c(y, sd[p + "norm.weight"], sd[p + "norm.bias"])
    if scheme is None:
        return O.block(x, sd, p)
    y = qconv(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], scheme)
    y = F.gelu(y)
    g = sd[p + "gamma"]
    y = qconv(y, sd[p + "mlp.fc2.weight"] * g.reshape(-1, 1, 1, 1), sd[p + "mlp.fc2.bias"] * g, scheme)
    return x + y


def features(img, sd, prefix, scheme):
    x = F.conv2d(img, sd[prefix + "stem.0.weight"], sd[prefix + "stem.0.bias"], stride=4)
    x = O.layer_norm_c(x, sd[prefix + "stem.1.weight"], sd[prefix + "stem.1.bias"])
    for i, depth in enumerate((2, 2, 6, 2)):
        sp = f"{prefix}stages.{i}."
        if i > 0:
            x = O.layer_norm_c(x, sd[sp + "downsample.0.weight"], sd[sp + "downsample.0.bias"])
            x = F.conv2d(x, sd[sp + "downsample.1.weight"], sd[sp + "downsample.1.bias"], stride=2)
        for j in range(depth):
            x = block(x, sd, f"{sp}blocks.{j}.", scheme if i >= 2 else None)
    return x


def forward(sd, cfg, img, meta, scheme):
    f = features(img, sd, "convnext_backbone.", scheme).flatten(1)
    m = O.metadata_branch(meta, sd, "metadata_branch.", "gelu", True)
    return O.fusion_head(torch.cat((f, m), 1), sd, "combined_head.", "gelu")


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    kind, cfg = CONFIGS["mm_pico"]
    img, meta, _ = synthetic_batch(n, seed=2)
    with torch.no_grad():
        for gamma in (1.0, 0.1):
            sd = seeded_state(kind, cfg, seed=3, gamma=gamma)
            ref = O.forward(kind, sd, cfg, img, meta)
            assert (forward(sd, cfg, img, meta, None) - ref).abs().max() < 1e-4
            for scheme in ("per_filter", "block_w", "block_wx"):
                out = forward(sd, cfg, img, meta, scheme)
                ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs()
                print(f"layer scale {gamma}: {scheme:10s} max|dscore| {ds.max().item():.3e}  rms {ds.pow(2).mean().sqrt().item():.3e}")
