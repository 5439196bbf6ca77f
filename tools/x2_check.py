"""GPU check of the split-operand mode: score / logit error against the fp32 oracle on the stress weights, per-stage
tap errors, and the time per batch.  usage: [N=1024] python tools/x2_check.py [precision ...]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
dev = torch.device("cuda:0")
kind, cfg = CONFIGS["mm_pico"]
N = int(os.environ.get("N", "1024"))
precs = sys.argv[1:] or ["f16x2", "f16"]
sd = seeded_state(kind, cfg, seed=3, gamma=float(os.environ.get("GAMMA", "1")))
img, meta, _ = synthetic_batch(N, seed=2)
taps = {}
torch.set_num_threads(16)
with torch.no_grad():
    ref = O.mm_convnext_forward(sd, cfg, img, meta, taps=taps)
for prec in precs:
    m = build_model(kind, cfg, sd, dev, prec)
    m.set_debug_taps(True)
    out = run_model(kind, m, img.to(dev), meta.to(dev)).cpu()
    line = []
    for t in ("stem", "stage0", "stage1", "stage2", "stage3"):
        got = m.read_tap(t).cpu()                                 # rows of the LAST forward chunk
        r = taps[t].permute(0, 2, 3, 1).reshape(-1, got.shape[-1])
        g = got.reshape(-1, got.shape[-1])
        rr = r[-g.shape[0]:]
        line.append(f"{t} {(g - rr).abs().max().item() / max(1.0, rr.abs().max().item()):.2e}")
    m.set_debug_taps(False)
    ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs()
    dl = (out - ref).abs()
    a, b = img.to(dev), meta.to(dev)
    for _ in range(5):
        run_model(kind, m, a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run_model(kind, m, a, b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{prec:6s}: N {N} max|dscore| {ds.max().item():.3e} rms {ds.pow(2).mean().sqrt().item():.2e} "
          f"max|dlogit| {dl.max().item():.3e} | taps(rel) {' '.join(line)} | {ms:.3f} ms per batch = {N / ms * 1e3:.0f} alerts/s",
          flush=True)
    m.set_profile(True)
    for _ in range(5):
        run_model(kind, m, a, b)
    prof = m.collect_profile()
    m.set_profile(False)
    print("   ", {k: round(v[0] / 5 * 1e3, 1) for k, v in prof.items() if v[1]}, "us per batch", flush=True)
