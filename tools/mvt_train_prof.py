"""Developer tool: a few mm_MaxViT training steps (for rocprofv3 --kernel-trace --stats).  usage: mvt_train_prof.py [batch] [steps]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
import bench
from btsbot_amd.train import Trainer
from btsbot_amd.synthetic import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mv = btsbot_amd.mm_MaxViT(bench.MAXVIT_CONFIG, precision="bf16")
bench.seeded_weights(mv)
mv = mv.to(dev).train()
img, meta, lab = [t.to(dev) for t in synthetic_batch(B, seed=70)]
tr = Trainer(mv, lr=1e-4, betas=(0.99, 0.99), pos_weight=1.0)
tr.step(img, meta, lab)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step(img, meta, lab)
torch.cuda.synchronize()
print(f"B={B}: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms per step, loss {float(loss):.4f}")
