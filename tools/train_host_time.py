"""Developer timing: how long the HOST needs to queue a training step (Trainer.step without any synchronisation)
against the step's GPU time -- if the two are close the GPU waits for Python / launch calls."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
from btsbot_amd.train import Trainer
from btsbot_amd.synthetic import synthetic_batch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = btsbot_amd.mm_ConvNeXt(bench.CONFIG, precision="bf16")
bench.seeded_weights(m)
m = m.to(dev).train()
img, meta, lab = synthetic_batch(B, seed=3)
img, meta, lab = img.to(dev), meta.to(dev), lab.to(dev)
tr = Trainer(m, lr=1e-4, betas=(0.99, 0.99), epochs=8, warmup_epochs=2)
for _ in range(5):
    tr.step(img, meta, lab)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    tr.step(img, meta, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host queues a step in {(t1 - t0) / n * 1e3:.3f} ms; GPU finishes one every {(t2 - t0) / n * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.step(img, meta, lab)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
