import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import CONFIGS, seeded_state, build_model, run_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
dev = torch.device('cuda:0')
kind, cfg = CONFIGS['mm_pico']
sd = seeded_state(kind, cfg, seed=3)
img, meta, _ = synthetic_batch(39, seed=2)
with torch.no_grad():
    taps = {}
    ref = O.mm_convnext_forward(sd, cfg, img, meta, taps=taps)
for prec in ('bf16', 'f16'):
    m = build_model(kind, cfg, sd, dev, prec)
    if os.environ.get('TAPS'): m.set_debug_taps(True)
    outs = []
    for rep in range(3):
        out = run_model(kind, m, img.to(dev), meta.to(dev)).cpu()
        outs.append(out)
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item()
        bad = ((torch.sigmoid(out) - torch.sigmoid(ref)).abs().flatten() > 3e-3).nonzero().flatten().tolist()
        print(os.environ.get('BTSBOT_AMD_S2P_DIAG'), prec, rep, 'dscore', ds, 'bad alerts', bad)
    print('   repeatable', torch.equal(outs[0], outs[1]), torch.equal(outs[1], outs[2]))
