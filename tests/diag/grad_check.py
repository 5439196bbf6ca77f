"""Developer diagnostic: per-tensor gradient error of the full backward vs autograd (fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from helpers import CONFIGS, seeded_state, build_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "convnext"
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
kind, cfg = CONFIGS[name]
import copy
cfg = copy.deepcopy(cfg)
for k in ("meta_dropout", "comb_dropout", "dropout"):
    if k in cfg: cfg[k] = 0.0
dev = torch.device("cuda:0")
sd = seeded_state(kind, cfg, seed=3)
B = 6
img, meta, labels = synthetic_batch(B, seed=4)
m = build_model(kind, cfg, sd, dev, prec).train()
if kind == "ConvNeXt": logits = m(input_data=img.to(dev))
else: logits = m(image_input=img.to(dev), metadata_input=meta.to(dev))
loss = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([2.0], device=dev))(logits, labels.to(dev).float().unsqueeze(1))
loss.backward()
ref = {k: v.clone() for k, v in sd.items()}
names = [k for k, _ in m.named_parameters()]
for k in names: ref[k].requires_grad_(True)
rl = O.bce_with_logits(O.forward(kind, ref, cfg, img, meta, training=True), labels.float().unsqueeze(1), 2.0)
rl.backward()
got = dict(m.named_parameters())
print("loss", loss.item(), rl.item())
for k in names:
    a, b = got[k].grad.cpu().double(), ref[k].grad.double()
    scale = max(b.abs().max().item(), 1e-9)
    err = (a - b).abs().max().item() / scale
    flag = "   <<<<" if err > 1e-3 else ""
    if flag or "--all" in sys.argv: print(f"{k:60s} rel {err:9.2e} scale {scale:9.2e}{flag}")
print("done")
