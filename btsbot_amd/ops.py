"""Thin tensor-level wrappers over the op-level C entry points (include/btsbot_hip.h,
``btsbot_op_*``).  Used by the op parity tests and the kernel micro-benchmarks; the model classes
call ``btsbot_forward`` instead.  HIP tensors only -- no CPU fallback."""
import ctypes as C

import torch

from . import _lib

_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def _p(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def _stream(t):
    if t.device.type != "cuda":
        raise RuntimeError("btsbot_amd.ops: tensors must live on a HIP device (no CPU fallback)")
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def gemm(x, w, bias, epi="bias", gamma=None, resid=None, precision="f32"):
    """epi in {'gelu','resid','bias'}; x [M,K], w [N,K] in the precision's dtype."""
    dt = _DT[precision]
    assert x.dtype == dt and w.dtype == dt and x.is_contiguous() and w.is_contiguous()
    M, K = x.shape
    N = w.shape[0]
    e = {"gelu": 0, "resid": 1, "bias": 2}[epi]
    out = torch.empty(M, N, dtype=dt if e == 0 else torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().btsbot_op_gemm(_lib.PRECISION[precision], e, _p(x), _p(w), _p(bias),
                                             _p(gamma), _p(resid), _p(out), M, N, K, _stream(x)),
                   "btsbot_op_gemm")
    return out


def dwconv_ln(x, w, bias, ln_w, ln_b, precision="f32"):
    """x [B,HW,HW,C] f32 NHWC; w [C,1,7,7] as PyTorch stores it."""
    B, HW, _, Cc = x.shape
    wt = w.reshape(Cc, 49).t().contiguous()
    out = torch.empty(B, HW, HW, Cc, dtype=_DT[precision], device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().btsbot_op_dwconv_ln(_lib.PRECISION[precision], _p(x), _p(wt), _p(bias),
                                                  _p(ln_w), _p(ln_b), _p(out), B, HW, Cc,
                                                  _stream(x)), "btsbot_op_dwconv_ln")
    return out


def stem(img, w, bias, ln_w, ln_b):
    B, C0 = img.shape[0], w.shape[0]
    out = torch.empty(B, 225, C0, dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.lib().btsbot_op_stem(_p(img), _p(w.contiguous()), _p(bias), _p(ln_w),
                                             _p(ln_b), _p(out), B, C0, _stream(img)),
                   "btsbot_op_stem")
    return out


def ln_patch(x, ln_w, ln_b, precision="f32"):
    B, HW, _, Cin = x.shape
    HO = HW // 2
    out = torch.empty(B * HO * HO, 4 * Cin, dtype=_DT[precision], device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().btsbot_op_ln_patch(_lib.PRECISION[precision], _p(x), _p(ln_w), _p(ln_b),
                                                 _p(out), B, HW, Cin, _stream(x)),
                   "btsbot_op_ln_patch")
    return out
