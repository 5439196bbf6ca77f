"""Measure max|dlogit| / max(1, max|logit|) and max|dscore| per wiring and precision (sets TOL_LOGIT_REL of the tests)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from helpers import CONFIGS, MV_CONFIGS, seeded_state, seeded_state_mv, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
from oracle import maxvit_oracle as MO
dev = torch.device("cuda:0")
torch.set_num_threads(16)
worst = {}
for name, (kind, cfg) in list(CONFIGS.items()) + list(MV_CONFIGS.items()):
    mv = name in MV_CONFIGS
    sd = (seeded_state_mv if mv else seeded_state)(kind, cfg, seed=3)
    img, meta, _ = synthetic_batch(5 if mv else 39, seed=2)
    with torch.no_grad():
        ref = (MO if mv else O).forward(kind, sd, cfg, img, meta)
    for prec in (["f32", "f16x2", "f16", "bf16"] + ([] if mv else ["fp8"])):
        m = build_model(kind, cfg, sd, dev, prec)
        out = run_model(kind, m, img.to(dev), meta.to(dev)).cpu()
        scale = max(1.0, ref.abs().max().item())
        dl = (out - ref).abs().max().item() / scale
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item()
        worst[prec] = max(worst.get(prec, 0.0), dl)
        print(f"{name:22s} {prec:6s} max|logit| {ref.abs().max().item():9.3f} dlogit/scale {dl:.3e} dscore {ds:.3e}", flush=True)
print("worst relative logit error per precision:", {k: f"{v:.2e}" for k, v in worst.items()})
