"""Developer tool (GPU box): one backward pass of mm_ConvNeXt-pico on seeded inputs; saves the gradient arena and the
parameter table so that two runs under different process-wide switches can be compared tensor by tensor.
usage: python tools/grad_ab.py save <file> [prec] [batch]      python tools/grad_ab.py cmp <file a> <file b>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

if sys.argv[1] == "save":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from helpers import CONFIGS, seeded_state, build_model
    from btsbot_amd.train import Trainer
    from btsbot_amd.synthetic import synthetic_batch
    prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 160
    kind, cfg0 = CONFIGS["mm_pico"]
    cfg = dict(cfg0, meta_dropout=0.0, comb_dropout=0.0)
    sd = seeded_state(kind, cfg, seed=3)
    dev = torch.device("cuda:0")
    img, meta, lab = synthetic_batch(B, seed=4)
    m = build_model(kind, cfg, sd, dev, prec).train()
    tr = Trainer(m, lr=1e-4)
    _l, g = tr.gradients(img.to(dev), meta.to(dev), lab.to(dev))
    torch.cuda.synchronize()
    table = [(r[0], int(r[1]), int(r[2])) for r in m._table_rows]
    np.savez(sys.argv[2], g=g.cpu().numpy(), table=np.array(table, dtype=object), allow_pickle=True)
    print("saved", sys.argv[2], g.numel())
else:
    a = np.load(sys.argv[2], allow_pickle=True)
    b = np.load(sys.argv[3], allow_pickle=True)
    ga, gb = a["g"], b["g"]
    d = np.abs(ga - gb)
    print(f"max |a-b| {d.max():.3e} at {d.argmax()}  max|a| {np.abs(ga).max():.3e}")
    tab = a["table"]
    if len(tab):
        rows = []
        for n, o, c in tab:
            seg = d[o:o + c]
            if seg.size:
                rows.append((seg.max() / max(np.abs(ga[o:o + c]).max(), 1e-30), seg.max(), np.abs(ga[o:o + c]).max(), n))
        for r in sorted(rows, reverse=True)[:12]:
            print(f"  rel {r[0]:.3e}  abs {r[1]:.3e}  max {r[2]:.3e}  {r[3]}")
