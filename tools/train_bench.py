"""Developer timing: full mm_ConvNeXt training steps (run under rocprofv3 for a kernel breakdown)."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
from btsbot_amd.train import Trainer
from btsbot_amd.synthetic import synthetic_batch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = btsbot_amd.mm_ConvNeXt(bench.CONFIG, precision=prec)
bench.seeded_weights(m)
m = m.to(dev).train()
img, meta, lab = synthetic_batch(B, seed=3)
img, meta, lab = img.to(dev), meta.to(dev), lab.to(dev)
tr = Trainer(m, lr=1e-4, betas=(0.99, 0.99), epochs=8, warmup_epochs=2)
for _ in range(5):
    tr.step(img, meta, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for _ in range(n):
    loss = tr.step(img, meta, lab)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B={B} {prec}: {dt*1e3:.3f} ms/step  {B/dt:.0f} alerts/s  loss {loss.item():.4f}")
