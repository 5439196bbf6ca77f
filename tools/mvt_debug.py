"""Developer aid: per-tensor gradient errors of the MaxViT branch training against autograd (tests/test_gpu_train.py)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_train as T
dev = torch.device("cuda:0")
dl, gerr, serr, after, sd = T._maxvit_branch_training_errors(dev, B=int(os.environ.get("B", "2")))
print("logit error", dl)
bad = [(k, v) for k, v in gerr.items() if not v <= 5e-4]
print(len(gerr), "tensors,", len(bad), "beyond 5e-4")
for k, v in list(gerr.items()):
    flag = "  <<<" if not v <= 5e-4 else ""
    if flag or os.environ.get("ALL"):
        print(f"{v:10.3e} {k}{flag}")
print("running-stat errors (worst 5):", sorted(serr.items(), key=lambda kv: -kv[1])[:5])
