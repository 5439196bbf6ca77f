"""CPU emulation of the 16-bit modes' roundings on the oracle: which rounding owns the score error?"""
import sys, torch, torch.nn.functional as F
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import CONFIGS, seeded_state
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
torch.set_num_threads(8)
kind, cfg = CONFIGS['mm_pico']
sd = seeded_state(kind, cfg, seed=3, gamma=float(os.environ.get('GAMMA', '1')))   # layer-scale magnitude
img, meta, _ = synthetic_batch(256, seed=2)
def rnd(t, dt, on):
    return t.to(dt).float() if on else t
def fwd(dt, r_xn=True, r_w=True, r_h=True, r_dwin=True, r_taps=True, r_stem=True, stages=(0,1,2,3)):
    arch = O.arch_of(cfg['model_kind']); depths = O.ARCHS[arch]['depths']; p = 'convnext_backbone.'
    x = F.conv2d(rnd(img, dt, r_stem), rnd(sd[p+'stem.0.weight'], dt, r_stem), sd[p+'stem.0.bias'], stride=4)
    x = O.layer_norm_c(x, sd[p+'stem.1.weight'], sd[p+'stem.1.bias'])
    for i, d in enumerate(depths):
        sp = f'{p}stages.{i}.'
        on = i in stages
        if i > 0:
            y = O.layer_norm_c(x, sd[sp+'downsample.0.weight'], sd[sp+'downsample.0.bias'])
            x = F.conv2d(rnd(y, dt, r_xn and on), rnd(sd[sp+'downsample.1.weight'], dt, r_w and on), sd[sp+'downsample.1.bias'], stride=2)
        for j in range(d):
            bp = f'{sp}blocks.{j}.'
            c = x.shape[1]
            dwin = rnd(x, dt, r_dwin and on and i < 2)          # stages 0/1: 16-bit map feeds the depthwise conv
            taps = rnd(sd[bp+'conv_dw.weight'], dt, r_taps and on and i < 2)
            y = F.conv2d(dwin, taps, sd[bp+'conv_dw.bias'], padding=3, groups=c)
            y = O.layer_norm_c(y, sd[bp+'norm.weight'], sd[bp+'norm.bias'])
            y = F.conv2d(rnd(y, dt, r_xn and on), rnd(sd[bp+'mlp.fc1.weight'], dt, r_w and on), sd[bp+'mlp.fc1.bias'])
            y = F.gelu(y)
            g = sd[bp+'gamma'].reshape(-1, 1, 1, 1)
            y = F.conv2d(rnd(y, dt, r_h and on), rnd(sd[bp+'mlp.fc2.weight'] * g, dt, r_w and on), sd[bp+'mlp.fc2.bias'] * sd[bp+'gamma'])
            x = x + y
    f = x.flatten(1)
    m = O.metadata_branch(meta, sd, 'metadata_branch.', 'gelu', True)
    return O.fusion_head(torch.cat((f, m), 1), sd, 'combined_head.', 'gelu')
with torch.no_grad():
    ref = fwd(torch.float16, False, False, False, False, False, False)
    ref2 = O.forward(kind, sd, cfg, img, meta)
    print('self check', (ref - ref2).abs().max().item())
    def rep(name, out):
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs()
        print(f'{name:40s} max|dscore| {ds.max().item():.2e}  rms {ds.pow(2).mean().sqrt().item():.2e}  max|dlogit| {(out-ref).abs().max().item():.2e}')
    for dt in (torch.float16, torch.bfloat16):
        print(dt)
        rep('all roundings', fwd(dt))
        rep('only xn (LN outputs -> fc1 / down)', fwd(dt, True, False, False, False, False, False))
        rep('only weights', fwd(dt, False, True, False, False, False, False))
        rep('only hidden h', fwd(dt, False, False, True, False, False, False))
        rep('only depthwise input map', fwd(dt, False, False, False, True, False, False))
        rep('only taps', fwd(dt, False, False, False, False, True, False))
        rep('only stem operands', fwd(dt, False, False, False, False, False, True))
        for st in range(4):
            rep(f'all roundings, stage {st} only', fwd(dt, True, True, True, True, True, st == 0, stages=(st,)))
