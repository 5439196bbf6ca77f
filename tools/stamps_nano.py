"""Developer diagnostic: in-kernel phase timeline (shader clock) of workgroup 0 of stage2p at convnext_nano's width."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd import _lib
from btsbot_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind, cfg = CONFIGS["mm_nano_ls"]
m = build_model(kind, cfg, seeded_state(kind, cfg, seed=3), dev, os.environ.get("PREC", "bf16"))
img, meta, _ = synthetic_batch(B, seed=2)
img, meta = img.to(dev), meta.to(dev)
for _ in range(3):
    run_model(kind, m, img, meta)
buf = torch.zeros(32 + 16384 + 64 + 2048, dtype=torch.int64, device=dev)
_lib.check(_lib.lib().btsbot_debug_stamps(m._handle.ptr, C.c_void_p(buf.data_ptr())), "stamps")
run_model(kind, m, img, meta)
torch.cuda.synchronize()
t = buf.cpu().tolist()
s2 = t[32 + 16384:32 + 16384 + 64]
print(f"stage2p<320> (workgroup 0): prologue +{s2[1] - s2[0]}")
for j in range(6):   # (blocks 6-7 share their slots with the downsample's stamps)
    b = 1 + 8 * j
    print(f"   block {j}: map->LDS +{s2[b+1]-s2[b]:6d}  depthwise +{s2[b+2]-s2[b+1]:6d}  LN +{s2[b+3]-s2[b+2]:6d}  "
          f"chunks 0-1 +{s2[b+4]-s2[b+3]:6d}  2-9 +{s2[b+6]-s2[b+4]:6d}  (block {s2[b+8] - s2[b]:7d})")
print(f"   downsample: LN +{s2[57]-s2[56]:6d}  conv +{s2[58]-s2[57]:6d}   whole kernel {s2[58]-s2[0]}")
