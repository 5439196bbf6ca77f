import sys, os, time, warnings
sys.path.insert(0, os.getcwd())
import torch, btsbot_amd, bench
from btsbot_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
for prec in ("bf16", "fp8"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.mm_ConvNeXt(bench.CONFIG, precision=prec)
    bench.seeded_weights(m); m = m.to(dev).eval()
    img, meta, _ = synthetic_batch(8192, seed=3); img, meta = img.to(dev), meta.to(dev)
    with torch.no_grad():
        for _ in range(5): m(image_input=img, metadata_input=meta)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): o = m(image_input=img, metadata_input=meta)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"{prec} B=8192: {dt*1e3:.3f} ms  {8192/dt/1e6:.3f} M alerts/s  tiles={os.environ.get('BTSBOT_AMD_S3_TILES','auto')}")
