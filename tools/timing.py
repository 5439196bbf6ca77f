"""Developer timing: forward at B alerts for each precision (HIP events on the current stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind, cfg = CONFIGS["mm_pico"]
sd = seeded_state(kind, cfg, seed=3)
img, meta, _ = synthetic_batch(B, seed=2)
img, meta = img.to(dev), meta.to(dev)
for prec in ("bf16", "f16", "f32"):
    m = build_model(kind, cfg, sd, dev, prec)
    for _ in range(3):
        run_model(kind, m, img, meta)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        run_model(kind, m, img, meta)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{prec}: B={B} {ms:.3f} ms/forward  {B / ms * 1e3:.0f} alerts/s")
