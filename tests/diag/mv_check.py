"""Developer check on the GPU box: MaxViT wirings vs the CPU oracle, with per-stage taps.
usage: python tests/diag/mv_check.py [B] [prec ...]"""
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import MM_MAXVIT, seeded_state_mv, build_model  # noqa: E402
from btsbot_amd.synthetic import synthetic_batch  # noqa: E402
from oracle import maxvit_oracle as MO  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
precs = sys.argv[2:] or ["f32", "bf16", "f16"]
dev = torch.device("cuda:0")
img, meta, _ = synthetic_batch(B, seed=2)
sd = seeded_state_mv("mm_MaxViT", MM_MAXVIT, seed=3)
taps = {}
t = time.time()
with torch.no_grad():
    ref = MO.mm_maxvit_forward(sd, MM_MAXVIT, img, meta, taps=taps)
print(f"oracle: {time.time() - t:.2f}s  logits {ref.flatten()[:4].tolist()}")
names = {"stem": "stem", "stage0": "s0b1", "stage1": "s1b1", "stage2": "s2b4", "stage3": "s3b1"}
for prec in precs:
    m = build_model("mm_MaxViT", MM_MAXVIT, sd, dev, prec)
    m.set_debug_taps(True)
    with torch.no_grad():
        out = m(image_input=img.to(dev), metadata_input=meta.to(dev)).cpu()
    for tap, key in names.items():
        got = m.read_tap(tap).cpu()
        want = taps[key].permute(0, 2, 3, 1).reshape(got.shape)
        err = (got - want).abs().max().item()
        print(f"  [{prec}] {tap:7s} max|d| {err:.3e}  (max|ref| {want.abs().max().item():.2f})")
    dl = (out - ref).abs().max().item()
    ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item()
    print(f"[{prec}] max|dlogit| {dl:.3e}  max|dscore| {ds:.3e}  logits {out.flatten()[:4].tolist()}")
