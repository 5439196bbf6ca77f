"""Developer timing: per-kernel average launch time (HIP events recorded by the library on the launch stream) of the
bf16 forward at B alerts, plus the score error against the oracle on 64 alerts (a wrong variant shows up at once).
BTSBOT_AMD_LIB=<path> times another build of the library (tools/build_variant.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import time
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
name = sys.argv[3] if len(sys.argv) > 3 else "mm_pico"
kind, cfg = CONFIGS[name]
sd = seeded_state(kind, cfg, seed=3)
img, meta, _ = synthetic_batch(B, seed=2)
with torch.no_grad():
    ref = O.forward(kind, sd, cfg, img[:64], meta[:64])
img, meta = img.to(dev), meta.to(dev)
m = build_model(kind, cfg, sd, dev, prec)
out = run_model(kind, m, img, meta)
err = (torch.sigmoid(out[:64].cpu()) - torch.sigmoid(ref)).abs().max().item()
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    for _ in range(20):
        run_model(kind, m, img, meta)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 100
e0.record()
for _ in range(n):
    run_model(kind, m, img, meta)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
m.set_profile(True)
for _ in range(n):
    run_model(kind, m, img, meta)
prof = m.collect_profile()
m.set_profile(False)
tag = os.environ.get("BTSBOT_AMD_LIB", "default")
parts = "  ".join(f"{k.replace('_kernel', '')} {1e3 * v / c:.1f}" for k, (v, c) in prof.items() if c)
print(f"[{os.path.basename(tag)}] {prec} B={B}: {ms * 1e3:.1f} us/forward ({B / ms * 1e3 / 1e6:.3f} M alerts/s), max|dscore| {err:.2e} | {parts}")
