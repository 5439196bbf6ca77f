"""Developer timing: mm_ConvNeXt with the convnext_nano table (dims 80/160/320/640: per-op schedule)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
import bench
from btsbot_amd.synthetic import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
kind = sys.argv[3] if len(sys.argv) > 3 else "convnext_nano.d1h_in1k"
dev = torch.device("cuda:0")
cfg = dict(bench.CONFIG, model_kind=kind)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = btsbot_amd.mm_ConvNeXt(cfg, precision=prec)
bench.seeded_weights(m)
m = m.to(dev).eval()
img, meta, _ = synthetic_batch(B, seed=3)
img, meta = img.to(dev), meta.to(dev)
with torch.no_grad():
    for _ in range(5):
        m(image_input=img, metadata_input=meta)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        out = m(image_input=img, metadata_input=meta)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
m.set_profile(True)
with torch.no_grad():
    m(image_input=img, metadata_input=meta)
prof = m.collect_profile()
print(f"{kind} B={B} {prec}: {dt*1e3:.3f} ms/step  {B/dt:.0f} alerts/s  finite={bool(torch.isfinite(out).all())}")
for k, (ms, cnt) in sorted(prof.items(), key=lambda x: -x[1][0]):
    if cnt:
        print(f"  {k:28s} {cnt:3d} launches {ms*1e3:8.1f} us")
# the same model through the streaming scorer (three batches in flight)
sc = btsbot_amd.ScoreStream(m, inputs_ready=True)
n = 60
for _ in range(2):
    for o in sc.map(((img, meta) for _ in range(n)), lag=n):
        pass
    torch.cuda.synchronize()
t0 = time.perf_counter()
for o in sc.map(((img, meta) for _ in range(n)), lag=n):
    pass
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"  ScoreStream(depth=3): {dt*1e3:.3f} ms/batch  {B/dt:.0f} alerts/s")
