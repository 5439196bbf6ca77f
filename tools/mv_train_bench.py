"""Developer timing: mm_MaxViT training steps alone (run under rocprofv3 for the per-kernel statistics).
usage: mv_train_bench.py [batch] [precision] [steps]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
import bench
from btsbot_amd.synthetic import synthetic_batch
from btsbot_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mv = btsbot_amd.mm_MaxViT(bench.MAXVIT_CONFIG, precision=prec)
bench.seeded_weights(mv)
mv = mv.to(dev).train()
img, meta, lab = synthetic_batch(B, seed=70)
img, meta, lab = img.to(dev), meta.to(dev), lab.to(dev)
tr = Trainer(mv, lr=1e-4, betas=(0.99, 0.99), pos_weight=1.0)
tr.step(img, meta, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step(img, meta, lab)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"mm_MaxViT train B={B} {prec}: {dt * 1e3:.2f} ms per step, {B / dt:.0f} alerts/s, loss {float(loss):.5f}")
