"""Developer micro-benchmark: the pointwise-conv GEMM shapes of ConvNeXt-pico at batch B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from btsbot_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[prec]
shapes = []
for P, Cc in ((225, 64), (49, 128), (9, 256), (1, 512)):
    shapes.append(("fc1", "gelu", B * P, 4 * Cc, Cc))
    shapes.append(("fc2", "resid", B * P, Cc, 4 * Cc))
for P, cin, cout in ((49, 64, 128), (9, 128, 256), (1, 256, 512)):
    shapes.append(("down", "bias", B * P, cout, 4 * cin))
for name, epi, M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    b = torch.randn(N, device=dev)
    g = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if epi == "resid" else None
    for _ in range(3):
        ops.gemm(x, w, b, epi, g, r, prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(x, w, b, epi, g, r, prec)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:5s} {epi:5s} M={M:7d} N={N:5d} K={K:5d}: {us:8.1f} us  {2 * M * N * K / us / 1e6:8.1f} TFLOP/s")
