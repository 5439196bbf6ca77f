"""Developer diagnostic: in-kernel phase timeline (shader clock) of workgroup 0 of the megakernels."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd import _lib
from btsbot_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind, cfg = CONFIGS["mm_pico"]
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
names = ["start", "stem/input", "b0 filters touched", "b0 DMA issued", "b0 depthwise done", "b0 DMA landed", "b0 MLP done",
         "b1 filters touched", "b1 DMA issued", "b1 depthwise done", "b1 DMA landed", "b1 MLP done", "ds LN done", "end"]
for base, tag in ((0, "stage0"), (16, "stage1")):
    print(tag, "total cycles", t[base + 13] - t[base])
    for i in range(1, 14):
        print(f"   {names[i]:22s} +{t[base + i] - t[base + i - 1]:8d}")

print("   stage0 b0 LN detail: reductions", t[14] - t[4], " barrier", t[15] - t[14], " combine + normalise + write", t[5] - t[15])
import numpy as np
for off, tag, n in ((32, "stage0", B), (32 + 8192, "stage1", (B + 1) // 2)):
    w = np.array(t[off:off + 2 * n]).reshape(n, 2)
    t0 = w[:, 0].min()
    dur = (w[:, 1] - w[:, 0]) / 100.0
    st = (w[:, 0] - t0) / 100.0
    print(tag, f"workgroups {n}: kernel span {(w[:,1].max()-t0)/100.0:.1f} us; WG duration us min/median/max "
          f"{dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f}; starts: {np.sum(st < 5)} within 5 us, "
          f"last start {st.max():.1f} us")
    print("   duration deciles", np.percentile(dur, [10, 30, 50, 70, 90]).round(1))
    print("   start deciles   ", np.percentile(st, [10, 30, 50, 70, 90]).round(1))

ls = t[32 + 8192 + 4096:32 + 8192 + 4096 + 16]
if any(ls):
    print("stage1 loop stamps (steps 6, 7):", [ls[i + 1] - ls[i] for i in range(15)])
s2 = t[32 + 16384:32 + 16384 + 64]
print("stage2p (workgroup 0) total cycles", s2[58] - s2[0])
print(f"   prologue (zero fill, x load)    +{s2[1] - s2[0]:8d}")
for j in range(6):
    b = 1 + 8 * j
    nxt = s2[1 + 8 * (j + 1)] if j < 5 else s2[56]
    print(f"   block {j}: map->LDS +{s2[b+1]-s2[b]:6d}  depthwise +{s2[b+2]-s2[b+1]:6d}  LN +{s2[b+3]-s2[b+2]:6d}  "
          f"chunks 0-1 +{s2[b+4]-s2[b+3]:6d}  2-7 +{s2[b+6]-s2[b+4]:6d}  (block {nxt - s2[b]:7d})")
print(f"   downsample: LN +{s2[57]-s2[56]:6d}  conv +{s2[58]-s2[57]:6d}")

s3 = t[32 + 16384 + 64:32 + 16384 + 64 + 16]
if any(s3):
    # the stage is four launches (s3_fc1, s3_fc2 per block): workgroup 0 of the LAST launch of each kernel leaves its stamps
    print("stage3, workgroup 0 of the last s3_fc1 launch: total cycles", s3[12] - s3[0], "(+ GELU and the h stores)")
    print(f"   rows + constants requested +{s3[9] - s3[0]:6d}  filter quarters + LayerNorm of 4 rows +{s3[10] - s3[9]:6d}  "
          f"bias + barrier +{s3[11] - s3[10]:6d}  MFMA loop +{s3[12] - s3[11]:6d}")
    print("stage3, workgroup 0 of the last s3_fc2 launch: total cycles", s3[14] - s3[1])
    print(f"   fragments requested + MFMA loop +{s3[13] - s3[1]:6d}  K slices through LDS, + x, stored +{s3[14] - s3[13]:6d}")

hs = t[32 + 16384 + 64 + 1500:32 + 16384 + 64 + 1500 + 10]
if any(hs):
    print("head16 (workgroup 0) total cycles", hs[8] - hs[0])
    for i, nm in enumerate(["loads issued", "feature LN", "meta in", "step 0", "step 1", "step 2", "step 3", "step 4"]):
        print(f"   {nm:14s} +{hs[i + 1] - hs[i]:7d}")
    d = t[32 + 16384 + 64 + 1500 + 10:32 + 16384 + 64 + 1500 + 15]   # slots 10, 12, 13, 14 (11 is unused)
    print("   inside meta fc2: entry -> fragments arrived + MFMA loop", d[2] - d[0], " barrier", d[3] - d[2], " epilogue", d[4] - d[3],
          " (entry at +", d[0] - hs[4], ")")
