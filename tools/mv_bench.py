"""Developer timing: mm_MaxViT inference steps alone (run under rocprofv3 for the per-kernel statistics / PMC passes of
BASELINE.json configs[3]).  usage: mv_bench.py [batch] [precision] [steps]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
import bench
from btsbot_amd.synthetic import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mv = btsbot_amd.mm_MaxViT(bench.MAXVIT_CONFIG, precision=prec)
bench.seeded_weights(mv)
mv = mv.to(dev).eval()
img, meta, _ = synthetic_batch(B, seed=50)
img, meta = img.to(dev), meta.to(dev)
with torch.no_grad():
    mv(image_input=img, metadata_input=meta)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = mv(image_input=img, metadata_input=meta)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"mm_MaxViT B={B} {prec}: {dt * 1e3:.2f} ms per forward, {B / dt:.0f} alerts/s, finite {bool(torch.isfinite(out).all())}")
