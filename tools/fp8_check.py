import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
dev = torch.device('cuda:0')
kind, cfg = CONFIGS['mm_pico']
for gamma in (1.0, 0.1):
    sd = seeded_state(kind, cfg, seed=3, gamma=gamma)
    img, meta, _ = synthetic_batch(256, seed=2)
    ref = O.forward(kind, sd, cfg, img, meta)
    for prec in ('bf16', 'fp8'):
        m = build_model(kind, cfg, sd, dev, prec)
        out = run_model(kind, m, img.to(dev), meta.to(dev)).cpu()
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs()
        print(f'gamma {gamma} {prec}: max|dscore| {ds.max().item():.3e} rms {ds.pow(2).mean().sqrt().item():.3e} finite {bool(torch.isfinite(out).all())}')
