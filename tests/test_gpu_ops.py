"""GPU parity of the op-level C entry points (include/btsbot_hip.h, ``btsbot_op_*``; btsbot_amd/ops.py): each
against the plain torch fp32 expression of the same operation (the ATen ops the reference dispatches to through
timm: conv2d / layer_norm / gelu; /root/reference/btsbot/architectures.py:108,132)."""
import pytest
import torch
import torch.nn.functional as F

from btsbot_amd import ops

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
# error of one rounding of the operands and of the result to the mode's operand type (relative to the result scale)
TOL = {"f32": 2e-5, "f16": 4e-3, "bf16": 3e-2}


def _rel(a, b):
    return (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-6)


@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(225 * 3, 256, 64), (49 * 5, 128, 512), (130, 1024, 256)])
def test_op_gemm_epilogues(cuda, prec, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g)).to(DT[prec]).to(cuda)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DT[prec]).to(cuda)
    bias = (0.1 * torch.randn(N, generator=g)).to(cuda)
    gamma = (1.0 + 0.1 * torch.randn(N, generator=g)).to(cuda)
    resid = torch.randn(M, N, generator=g).to(cuda)
    acc = x.float() @ w.float().t() + bias
    assert _rel(ops.gemm(x, w, bias, "bias", precision=prec), acc) <= TOL[prec]
    assert _rel(ops.gemm(x, w, bias, "gelu", precision=prec), F.gelu(acc)) <= TOL[prec]
    out = ops.gemm(x, w, bias, "resid", gamma=gamma, resid=resid, precision=prec)
    assert _rel(out, resid + gamma * acc) <= TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("hw,c", [(15, 64), (7, 128), (3, 256), (1, 512), (15, 80)])
def test_op_dwconv_ln(cuda, prec, hw, c):
    g = torch.Generator().manual_seed(hw * c)
    x = torch.randn(3, hw, hw, c, generator=g)
    w = torch.randn(c, 1, 7, 7, generator=g) / 7.0
    b, lw, lb = (0.1 * torch.randn(c, generator=g) for _ in range(3))
    lw = lw + 1.0
    y = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=3, groups=c).permute(0, 2, 3, 1)
    ref = F.layer_norm(y, (c,), lw, lb, 1e-6)
    out = ops.dwconv_ln(x.to(cuda), w.to(cuda), b.to(cuda), lw.to(cuda), lb.to(cuda), precision=prec)
    assert out.dtype == DT[prec]
    assert _rel(out.cpu(), ref) <= TOL[prec]


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("c", [64, 80])
def test_op_dwconv_ln_15x15_at_the_full_batch(cuda, prec, c):
    """BASELINE.json's batch (1024 alerts; an odd count too: C = 80 runs two alerts per workgroup): the matrix-pipe
    depthwise + LayerNorm of the 15x15 maps (dw15.hip) against the fp32 per-tap kernel on the same device tensors --
    every alert, not only the first workgroups'."""
    g = torch.Generator().manual_seed(c)
    for B in (1024, 1023):
        x = torch.randn(B, 15, 15, c, generator=g).to(cuda)
        w = (torch.randn(c, 1, 7, 7, generator=g) / 7.0).to(cuda)
        b, lw, lb = (0.1 * torch.randn(c, generator=g).to(cuda) for _ in range(3))
        lw = lw + 1.0
        ref = ops.dwconv_ln(x, w, b, lw, lb, precision="f32")
        out = ops.dwconv_ln(x, w, b, lw, lb, precision=prec)
        assert out.dtype == DT[prec] and out.shape == ref.shape
        err = (out.float() - ref).abs().amax(dim=(1, 2, 3)) / ref.abs().amax()
        assert float(err.max()) <= TOL[prec], (B, int(err.argmax()), float(err.max()))


@pytest.mark.parametrize("c0", [64, 80])
def test_op_stem(cuda, c0):
    g = torch.Generator().manual_seed(c0)
    img = torch.randn(4, 3, 63, 63, generator=g)
    w = torch.randn(c0, 3, 4, 4, generator=g) / 7.0
    b, lw, lb = (0.1 * torch.randn(c0, generator=g) for _ in range(3))
    lw = lw + 1.0
    y = F.conv2d(img, w, b, stride=4).permute(0, 2, 3, 1)
    ref = F.layer_norm(y, (c0,), lw, lb, 1e-6).reshape(4, 225, c0)
    out = ops.stem(img.to(cuda), w.to(cuda), b.to(cuda), lw.to(cuda), lb.to(cuda))
    assert _rel(out.cpu(), ref) <= 2e-5


@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("hw,cin", [(15, 64), (7, 128), (3, 256)])
def test_op_ln_patch(cuda, prec, hw, cin):
    g = torch.Generator().manual_seed(hw + cin)
    x = torch.randn(2, hw, hw, cin, generator=g)
    lw = 1.0 + 0.1 * torch.randn(cin, generator=g)
    lb = 0.1 * torch.randn(cin, generator=g)
    y = F.layer_norm(x, (cin,), lw, lb, 1e-6)
    ho = hw // 2
    ref = torch.stack([y[:, ky:2 * ho:2, kx:2 * ho:2, :] for ky in range(2) for kx in range(2)], dim=3)
    ref = ref.reshape(2 * ho * ho, 4 * cin)          # k = (ky * 2 + kx) * cin + c
    out = ops.ln_patch(x.to(cuda), lw.to(cuda), lb.to(cuda), precision=prec)
    assert _rel(out.cpu(), ref) <= TOL[prec]
