"""Developer diagnostic: in-kernel phase timeline (shader clock, workgroup 0) of stage2p_kernel's keeping form inside a
training forward.  usage: stamps_train.py [B] [prec]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd, bench
from btsbot_amd import _lib
from btsbot_amd.train import Trainer
from btsbot_amd.synthetic import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
m = btsbot_amd.mm_ConvNeXt(bench.CONFIG, precision=prec)
bench.seeded_weights(m)
m = m.to(dev).train()
img, meta, lab = [t.to(dev) for t in synthetic_batch(B, seed=3)]
tr = Trainer(m, lr=1e-4)
for _ in range(3):
    tr.step(img, meta, lab)
buf = torch.zeros(32 + 16384 + 64 + 2048, dtype=torch.int64, device=dev)
_lib.check(_lib.lib().btsbot_debug_stamps(m._handle.ptr, C.c_void_p(buf.data_ptr())), "stamps")
tr.step(img, meta, lab)
torch.cuda.synchronize()
_lib.check(_lib.lib().btsbot_debug_stamps(m._handle.ptr, C.c_void_p(0)), "stamps")
t = buf.cpu().tolist()
s2 = t[32 + 16384:32 + 16384 + 64]
print("stage2p keeping form (workgroup 0) total cycles", s2[58] - s2[0])
print(f"   prologue (zero fill, x load)    +{s2[1] - s2[0]:8d}")
for j in range(6):
    b = 1 + 8 * j
    nxt = s2[1 + 8 * (j + 1)] if j < 5 else s2[56]
    print(f"   block {j}: map->LDS +{s2[b+1]-s2[b]:6d}  depthwise +{s2[b+2]-s2[b+1]:6d}  LN +{s2[b+3]-s2[b+2]:6d}  "
          f"chunks 0-1 +{s2[b+4]-s2[b+3]:6d}  2-7 +{s2[b+6]-s2[b+4]:6d}  (block {nxt - s2[b]:7d})")
print(f"   downsample: LN +{s2[57]-s2[56]:6d}  conv +{s2[58]-s2[57]:6d}")
