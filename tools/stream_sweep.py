"""Developer sweep: alerts/s of the streaming scorer over (batches in flight, alerts per library chunk) at a given
call size and precision.  usage: stream_sweep.py [precision] [alerts per call]"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
import bench
from btsbot_amd.synthetic import synthetic_batch
prec = sys.argv[1] if len(sys.argv) > 1 else "fp8"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda:0")
img, meta, _ = synthetic_batch(B, seed=3)
img, meta = img.to(dev), meta.to(dev)
for chunk in (1024, 2048, 4096):
    for depth in (2, 3, 4):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = btsbot_amd.mm_ConvNeXt(bench.CONFIG, precision=prec)
        bench.seeded_weights(m)
        m = m.to(dev).eval()
        m._max_chunk = chunk
        sc = btsbot_amd.ScoreStream(m, depth=depth, inputs_ready=True)
        for r in sc.models:
            r._max_chunk = chunk
        n = max(12, 98304 // B)
        for _ in range(2):
            for o in sc.map(((img, meta) for _ in range(n)), lag=n):
                pass
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in sc.map(((img, meta) for _ in range(n)), lag=n):
            pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{prec} B={B} chunk {chunk} depth {depth}: {B * n / dt / 1e6:.3f} M alerts/s", flush=True)
        del sc, m
