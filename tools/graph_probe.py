"""Developer probe: the whole training step captured into one HIP graph (torch.cuda.graph) against the eager step:
CPU enqueue time per step, wall time per step, and the graphed replay."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
from btsbot_amd.train import Trainer
from btsbot_amd.synthetic import synthetic_batch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
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
for _ in range(n):
    loss = tr.step(img, meta, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: enqueue {(t1 - t0) / n * 1e3:.3f} ms/step, wall {(t2 - t0) / n * 1e3:.3f} ms/step  loss {loss.item():.4f}", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        tr.step(img, meta, lab)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gl = tr.step(img, meta, lab)
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph: wall {(t2 - t0) / n * 1e3:.3f} ms/step  loss {gl.item():.4f}", flush=True)
