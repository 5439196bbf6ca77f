"""Developer sweep: convnext_nano through the streaming scorer at several depths / call sizes."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
import bench
from btsbot_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
cfg = dict(bench.CONFIG, model_kind="convnext_nano.d1h_in1k")
for B in (1024, 2048, 4096):
    img, meta, _ = synthetic_batch(B, seed=3)
    img, meta = img.to(dev), meta.to(dev)
    for depth in (2, 3, 4, 6):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = btsbot_amd.mm_ConvNeXt(cfg, precision="bf16")
        bench.seeded_weights(m)
        m = m.to(dev).eval()
        sc = btsbot_amd.ScoreStream(m, depth=depth, inputs_ready=True)
        n = max(12, 61440 // B)
        for _ in range(2):
            for o in sc.map(((img, meta) for _ in range(n)), lag=n):
                pass
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in sc.map(((img, meta) for _ in range(n)), lag=n):
            pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"nano B={B} depth {depth}: {dt*1e3:.3f} ms/batch  {B/dt/1e6:.3f} M alerts/s", flush=True)
        del sc, m
