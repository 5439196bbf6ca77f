"""Developer probe: the split mode's score error on five weight seeds x 1024 alerts against the fp32 oracle (the
north-star test's inputs), and its forward time -- run with BTSBOT_AMD_X2_TAIL_F16=1 to see what plain f16 operands in
stages 2-3 cost in error and buy in time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
dev = torch.device("cuda:0")
kind, cfg = CONFIGS["mm_pico"]
worst = []
for seed in (3, 11, 12, 13, 14):
    sd = seeded_state(kind, cfg, seed=seed)
    img, meta, _ = synthetic_batch(1024, seed=20 + seed)
    with torch.no_grad():
        ref = O.forward(kind, sd, cfg, img, meta)
    m = build_model(kind, cfg, sd, dev, "f16x2")
    di, dm = img.to(dev), meta.to(dev)
    out = run_model(kind, m, di, dm).cpu()
    ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs()
    worst.append(ds.max().item())
    print(f"seed {seed}: max|dscore| {worst[-1]:.3e} rms {ds.pow(2).mean().sqrt().item():.3e}", flush=True)
    if seed == 14:
        for _ in range(20):
            run_model(kind, m, di, dm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            run_model(kind, m, di, dm)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f"forward {dt*1e3:.3f} ms per 1024 alerts = {1024/dt/1e6:.2f} M alerts/s")
print("worst", max(worst))
