"""Which operands of which stages must be split (hi + lo f16) for |dscore| <= 1e-4?  CPU emulation on the oracle.
scheme = {stage: set of rounded operand kinds}, kinds: xn (LN outputs), w (1x1 / downsample filters), h (hidden),
dwin (depthwise input map), taps (depthwise filter), stem."""
import sys, os, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import CONFIGS, seeded_state
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
torch.set_num_threads(8)
kind, cfg = CONFIGS['mm_pico']
sd = seeded_state(kind, cfg, seed=3, gamma=float(os.environ.get('GAMMA', '1')))
N = int(os.environ.get('N', '1024'))
img, meta, _ = synthetic_batch(N, seed=2)
dt = torch.float16
def rnd(t, on):
    return t.to(dt).float() if on else t
def fwd(scheme, stem=True):
    arch = O.arch_of(cfg['model_kind']); depths = O.ARCHS[arch]['depths']; p = 'convnext_backbone.'
    x = F.conv2d(rnd(img, stem), rnd(sd[p+'stem.0.weight'], stem), sd[p+'stem.0.bias'], stride=4)
    x = O.layer_norm_c(x, sd[p+'stem.1.weight'], sd[p+'stem.1.bias'])
    for i, d in enumerate(depths):
        sp = f'{p}stages.{i}.'
        k = scheme.get(i, set())
        if i > 0:
            kd = scheme.get(('down', i), k)
            y = O.layer_norm_c(x, sd[sp+'downsample.0.weight'], sd[sp+'downsample.0.bias'])
            x = F.conv2d(rnd(y, 'xn' in kd), rnd(sd[sp+'downsample.1.weight'], 'w' in kd), sd[sp+'downsample.1.bias'], stride=2)
        for j in range(d):
            bp = f'{sp}blocks.{j}.'
            c = x.shape[1]
            y = F.conv2d(rnd(x, 'dwin' in k and i < 2), rnd(sd[bp+'conv_dw.weight'], 'taps' in k and i < 2), sd[bp+'conv_dw.bias'], padding=3, groups=c)
            y = O.layer_norm_c(y, sd[bp+'norm.weight'], sd[bp+'norm.bias'])
            y = F.conv2d(rnd(y, 'xn' in k), rnd(sd[bp+'mlp.fc1.weight'], 'w' in k), sd[bp+'mlp.fc1.bias'])
            y = F.gelu(y)
            g = sd[bp+'gamma'].reshape(-1, 1, 1, 1)
            y = F.conv2d(rnd(y, 'h' in k), rnd(sd[bp+'mlp.fc2.weight'] * g, 'w' in k), sd[bp+'mlp.fc2.bias'] * sd[bp+'gamma'])
            x = x + y
    f = x.flatten(1)
    m = O.metadata_branch(meta, sd, 'metadata_branch.', 'gelu', True)
    return O.fusion_head(torch.cat((f, m), 1), sd, 'combined_head.', 'gelu')
ALL = {'xn', 'w', 'h', 'dwin', 'taps'}
with torch.no_grad():
    ref = fwd({}, stem=False)
    def rep(name, out):
        ds = (torch.sigmoid(out) - torch.sigmoid(ref)).abs()
        print(f'{name:70s} max {ds.max().item():.2e}  rms {ds.pow(2).mean().sqrt().item():.2e}', flush=True)
    rep('plain f16 everywhere', fwd({i: ALL for i in range(4)}))
    rep('only stem rounded', fwd({}))
    rep('stages 0,1 plain f16; 2,3 exact', fwd({0: ALL, 1: ALL, ('down', 2): set(), ('down', 3): set()}))
    rep('stages 0,1: dwin only; rest exact', fwd({0: {'dwin'}, 1: {'dwin'}, ('down',1): set(), ('down',2): set()}))
    rep('stages 0,1: dwin+taps; rest exact', fwd({0: {'dwin','taps'}, 1: {'dwin','taps'}, ('down',1): set(), ('down',2): set()}))
    rep('stages 0,1: xn,h,dwin (weights+taps split)', fwd({0: {'xn','h','dwin'}, 1: {'xn','h','dwin'}, ('down',1): set(), ('down',2): set()}))
    rep('stages 0,1: xn,dwin (w,taps,h split)', fwd({0: {'xn','dwin'}, 1: {'xn','dwin'}, ('down',1): set(), ('down',2): set()}))
    rep('stages 0,1: h,dwin (w,taps,xn split)', fwd({0: {'h','dwin'}, 1: {'h','dwin'}, ('down',1): set(), ('down',2): set()}))
    rep('stage 0: all f16, rest exact', fwd({0: ALL, ('down',1): set()}))
    rep('stage 0: xn,h,dwin, rest exact', fwd({0: {'xn','h','dwin'}, ('down',1): set()}))
    rep('stage 0: dwin,taps, rest exact', fwd({0: {'dwin','taps'}, ('down',1): set()}))
    rep('stage 1: all f16 (down f16 too), rest exact', fwd({1: ALL}))
    rep('stage 1: xn,h,dwin, rest exact', fwd({1: {'xn','h','dwin'}, ('down',1): set()}))
    rep('stage 2: xn only rounded', fwd({2: {'xn'}, ('down',2): set()}))
    rep('stage 2: h only rounded', fwd({2: {'h'}, ('down',2): set()}))
    rep('stage 2: w only rounded', fwd({2: {'w'}, ('down',2): set()}))
    rep('stage 3: xn only', fwd({3: {'xn'}, ('down',3): set()}))
    rep('stage 3: h only', fwd({3: {'h'}, ('down',3): set()}))
    rep('stage 3: w only', fwd({3: {'w'}, ('down',3): set()}))
    rep('downsamples only (xn+w)', fwd({('down',1): {'xn','w'}, ('down',2): {'xn','w'}, ('down',3): {'xn','w'}}))
