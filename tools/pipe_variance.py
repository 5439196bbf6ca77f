"""How stable is the two-stream overlap from process to process?  (stream -> hardware-queue mapping is the runtime's)"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
from bench import CONFIG, seeded_weights
from btsbot_amd.synthetic import synthetic_batch
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = btsbot_amd.mm_ConvNeXt(CONFIG, precision="bf16")
seeded_weights(m)
m = m.to(dev).eval()
img, meta, _ = synthetic_batch(1024, seed=3)
img, meta = img.to(dev), meta.to(dev)
def timeit(fn, n=200):
    fn(20); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
def serial(n):
    with torch.no_grad():
        for _ in range(n):
            m(image_input=img, metadata_input=meta)
ts = timeit(serial)
sc = btsbot_amd.ScoreStream(m, depth=2, inputs_ready=True)
if mode == "prio":
    sc.streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
def piped(n):
    for _ in sc.map((img, meta) for _ in range(n)):
        pass
tp = [timeit(piped) for _ in range(3)]
print(f"{mode}: serial {ts:.4f} ms, pipelined " + " ".join(f"{t:.4f}" for t in tp))
