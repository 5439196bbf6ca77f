"""Developer probe: which part of bench.py's flow in front of its training leg slows the two-stream backward down?"""
import sys, os, time, warnings, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
from btsbot_amd.train import Trainer
from btsbot_amd.synthetic import synthetic_batch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
dev = torch.device("cuda:0")
img, meta, lab = (t.to(dev) for t in synthetic_batch(1024, seed=3))


def mk(train):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.mm_ConvNeXt(bench.CONFIG, precision="bf16")
    bench.seeded_weights(m)
    m = m.to(dev)
    return m.train() if train else m.eval()


if mode.startswith("streams"):
    ss = [torch.cuda.Stream(device=dev) for _ in range(int(mode[7:]))]
    for x in ss:
        x.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
elif mode == "replicas":
    ms = [mk(False) for _ in range(3)]
    with torch.no_grad():
        for x in ms:
            x(image_input=img, metadata_input=meta)
    torch.cuda.synchronize()
elif mode != "none":
    model = mk(False)
    depth = 2 if mode == "depth2" else 3
    sc = btsbot_amd.ScoreStream(model, depth=depth, inputs_ready=True)
    if mode != "norun":
        for o in sc.map(((img, meta) for _ in range(20)), lag=20):
            pass
    torch.cuda.synchronize()
    if mode == "del":
        del sc, o
        gc.collect()
        torch.cuda.synchronize()
tm = mk(True)
tr = Trainer(tm, lr=1e-4, betas=(0.99, 0.99), epochs=8, warmup_epochs=2)
for _ in range(5):
    tr.step(img, meta, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 40
for _ in range(n):
    tr.step(img, meta, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{mode}: host {(t1 - t0) / n * 1e3:.3f} ms; GPU {(t2 - t0) / n * 1e3:.3f} ms per step", flush=True)
