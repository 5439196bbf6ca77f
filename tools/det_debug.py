"""Developer check: which gradient tensors differ between identical passes in the deterministic mode.
usage: det_debug.py [prec] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["BTSBOT_AMD_DETERMINISTIC"] = "1"
import torch
from helpers import CONFIGS, seeded_state, build_model
from btsbot_amd.synthetic import synthetic_batch
from btsbot_amd.train import Trainer
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cuda = torch.device("cuda:0")
kind, cfg0 = CONFIGS["mm_pico"]
cfg = dict(cfg0, meta_dropout=0.0, comb_dropout=0.0)
sd = seeded_state(kind, cfg, seed=3)
img, meta, lab = [t.to(cuda) for t in synthetic_batch(B, seed=4)]
m = build_model(kind, cfg, sd, cuda, prec).train()
tr = Trainer(m, lr=1e-4)
out = []
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for _ in range(NP):
    _l, g = tr.gradients(img, meta, lab)
    torch.cuda.synchronize()
    out.append(g.clone())
names = [(k, p) for k, p in m.named_parameters()]
off = 0
info = m._handle.param_table() if hasattr(m._handle, "param_table") else None
for i in range(1, NP):
    d = (out[i] - out[0]).abs()
    print(f"pass {i} vs 0: max |diff| {d.max().item():.3e}, {int((d > 0).sum())} entries differ of {d.numel()}")
    if d.max().item() > 0:
        idx = torch.nonzero(d > 0).flatten()
        print("   first / last differing arena index:", int(idx[0]), int(idx[-1]))
# map arena indices to parameter names through the views the module holds
arena = m._arena if hasattr(m, "_arena") else None
if arena is not None:
    bad = [i for i in range(1, NP) if not torch.equal(out[i], out[0])]
    d = (out[bad[0] if bad else 1] - out[0]).abs()
    base = arena.data_ptr()
    for k, p in names:
        o = (p.data_ptr() - base) // 4
        dd = d[o:o + p.numel()]
        if dd.numel() and dd.max().item() > 0:
            print(f"   {k}: max {dd.max().item():.3e} ({int((dd > 0).sum())} of {p.numel()})")
