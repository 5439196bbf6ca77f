"""Developer diagnostic: in-kernel phase timeline (shader clock) of workgroup 0's first unit of mv_part_kernel."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, warnings
import btsbot_amd
import bench
from btsbot_amd import _lib
from btsbot_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mv = btsbot_amd.mm_MaxViT(bench.MAXVIT_CONFIG, precision=os.environ.get("PREC", "bf16"))
bench.seeded_weights(mv)
mv = mv.to(dev).eval()
img, meta, _ = synthetic_batch(B, seed=50)
img, meta = img.to(dev), meta.to(dev)
def run():
    with torch.no_grad():
        return mv(image_input=img, metadata_input=meta)
for _ in range(2):
    run()
buf = torch.zeros(24000, dtype=torch.int64, device=dev)
_lib.check(_lib.lib().btsbot_debug_stamps(mv._handle.ptr, C.c_void_p(buf.data_ptr())), "stamps")
run()
torch.cuda.synchronize()
t = buf.cpu().tolist()
names = ["rows requested", "LN1", "qkv", "proj frag + bias requested, barrier", "attention", "barrier", "proj", "LN2",
         "fc1+GELU step 0", "barrier", "fc2 step 0", "fc1+GELU step 1", "barrier", "fc2 step 1", "remaining steps", "rows stored"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
for base, tag in ((20000, "C=256"), (20032, "C=128")):
    s = t[base:base + 32]
    if not any(s):
        continue
    print(tag, "unit total, shader clocks:", s[16] - s[0])
    for i in range(16):
        if s[idx[i + 1]] and s[idx[i]]:
            print(f"   {names[i]:40s} +{s[idx[i + 1]] - s[idx[i]]:6d}")
