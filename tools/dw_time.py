"""Developer timing of the depthwise + LayerNorm op (btsbot_op_dwconv_ln) at B alerts: us per launch.
BTSBOT_AMD_NO_DW15=1 times the per-tap kernel for the 15x15 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from btsbot_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
for hw, c in ((15, 64), (15, 80), (7, 128), (7, 160)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, hw, hw, c, generator=g).to(dev)
    w = (torch.randn(c, 1, 7, 7, generator=g) / 7.0).to(dev)
    b, lw, lb = (0.1 * torch.randn(c, generator=g).to(dev) for _ in range(3))
    for _ in range(5):
        ops.dwconv_ln(x, w, b, lw, lb, precision="bf16")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.dwconv_ln(x, w, b, lw, lb, precision="bf16")
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = B * hw * hw * c * 6 / 1e6
    print(f"hw {hw} C {c} B {B}: {us:.1f} us  ({mb / us:.2f} TB/s of x + xn)", flush=True)
