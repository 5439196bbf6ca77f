"""Developer diagnostic: per-stage error of the HIP path against the oracle, all precisions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 39
img, meta, _ = synthetic_batch(B, seed=2)
for name in (sys.argv[2:] or ["mm_pico"]):
    kind, cfg = CONFIGS[name]
    sd = seeded_state(kind, cfg, seed=3)
    taps = {}
    with torch.no_grad():
        if kind == "mm_ConvNeXt":
            ref = O.mm_convnext_forward(sd, cfg, img, meta, taps=taps)
        else:
            ref = O.forward(kind, sd, cfg, img, meta)
    for prec in ("f32", "bf16", "f16"):
        m = build_model(kind, cfg, sd, dev, prec)
        if kind != "um_nn":
            m.set_debug_taps(True)
        out = run_model(kind, m, img.to(dev), meta.to(dev)).cpu()
        torch.cuda.synchronize()
        dl = (out - ref).abs().max().item()
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item()
        print(f"{name} {prec}: max|dlogit|={dl:.3e} max|dscore|={ds:.3e} logits[:4]={out.flatten()[:4].tolist()} ref={ref.flatten()[:4].tolist()}")
        if kind == "mm_ConvNeXt":
            for t in ("stem", "stage0", "stage1", "stage2", "stage3"):
                g = m.read_tap(t).cpu()
                r = taps[t].permute(0, 2, 3, 1).reshape(g.shape)
                print(f"   {t}: max|d|={(g - r).abs().max().item():.3e} ref_rms={r.pow(2).mean().sqrt().item():.3e}")
