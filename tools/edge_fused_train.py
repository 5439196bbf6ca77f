"""Developer check: gradients of the fused (mlp_bwd_kernel) and the unfused 16-bit training schedules against autograd
through the fp32 oracle at a batch size of choice.  usage: edge_fused_train.py [prec] [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CONFIGS, seeded_state, build_model
from btsbot_amd.synthetic import synthetic_batch
from oracle import convnext_oracle as O
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cuda = torch.device("cuda:0")
kind, cfg = CONFIGS["mm_pico"]
cfg = dict(cfg, meta_dropout=0.0, comb_dropout=0.0)
sd = seeded_state(kind, cfg, seed=3)
img, meta, labels = synthetic_batch(B, seed=4)


def grads(p):
    m = build_model(kind, cfg, sd, cuda, p).train()
    logits = m(image_input=img.to(cuda), metadata_input=meta.to(cuda))
    loss = torch.nn.BCEWithLogitsLoss()(logits, labels.to(cuda).float().unsqueeze(1))
    loss.backward()
    return {k: q.grad.detach().cpu().double() for k, q in m.named_parameters()}


torch.set_num_threads(16)
ref = {k: v.clone() for k, v in sd.items()}
names = list(grads("f32").keys())
for k in names:
    ref[k].requires_grad_(True)
O.bce_with_logits(O.forward(kind, ref, cfg, img, meta, training=True), labels.float().unsqueeze(1), 1.0).backward()
want = {k: ref[k].grad.double() for k in names}


def worst(a, b):
    return max(((a[k] - b[k]).abs().max().item() / max(b[k].abs().max().item(), 1e-12), k) for k in names)


os.environ.pop("BTSBOT_AMD_NO_MLP_BWD", None)
fused = grads(prec)
os.environ["BTSBOT_AMD_NO_MLP_BWD"] = "1"
plain = grads(prec)
print(f"{prec} B={B}: fused vs oracle %.2e (%s)" % worst(fused, want))
print(f"{prec} B={B}: unfused vs oracle %.2e (%s)" % worst(plain, want))
print(f"{prec} B={B}: fused vs unfused %.2e (%s)" % worst(fused, plain))
