"""Generate the committed golden vectors.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

For each wiring it runs the REFERENCE'S OWN wrapper class (/root/reference/btsbot/architectures.py,
imported by file path with oracle/timm_standin.py standing in for the absent timm package) on
fixed inputs with seeded weights and stores inputs + logits:

  example8.npz      8 of the 39 bundled example alerts (4 per label; float32 NCHW exactly as
                    inference_example.py:62-64 prepares them), their 25 metadata columns
                    (inference_example.py:53-58), labels and the csv's `expected_scores` column
  ref_logits.npz    logits of the reference wrappers for every config in tests/helpers.py on
                    (a) example8 and (b) synthetic_batch(6, seed=2); seeded weights
                    (oracle.random_state_dict(seed=3)) are regenerated at test time, a float64
                    checksum of every state dict is stored to detect RNG drift
  ref_logits_maxvit.npz  the same for the reference's MaxViT / mm_MaxViT wrappers around the
                    maxvit_tiny_rw_224 stand-in (first 4 example alerts + synthetic_batch(3, seed=2))
  lr_sequences.json torch's own SequentialLR sequences for (warmup, epochs) in {(0,6),(2,8)}
  adamw_bce.npz     3 AdamW steps (torch.optim.AdamW) and BCEWithLogitsLoss(pos_weight) values

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import json
import os
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import CONFIGS, MV_CONFIGS, seeded_state, seeded_state_mv  # noqa: E402
from btsbot_amd.synthetic import METADATA_COLS, synthetic_batch  # noqa: E402
from oracle import convnext_oracle as O, maxvit_oracle as MO, timm_standin  # noqa: E402

REF = "/root/reference/btsbot"


def load_reference_architectures():
    timm_standin.install()
    spec = importlib.util.spec_from_file_location("ref_architectures", f"{REF}/architectures.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def example8():
    cand = pd.read_csv(f"{REF}/example_data/usage_candidates.csv", index_col=None)
    trip = np.load(f"{REF}/example_data/usage_triplets.npy", mmap_mode="r").astype(np.float32)
    trip = np.ascontiguousarray(np.transpose(trip, (0, 3, 1, 2)))
    pos = np.where(cand["label"].values == 1)[0][:4]
    neg = np.where(cand["label"].values == 0)[0][:4]
    idx = np.concatenate([pos, neg])
    return dict(index=idx, triplets=trip[idx],
                metadata=cand[METADATA_COLS].values.astype(np.float32)[idx],
                labels=cand["label"].values[idx].astype(np.int64),
                expected_scores=cand["expected_scores"].values[idx].astype(np.float64))


def checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def call(kind, model, img, meta):
    with torch.no_grad():
        if kind in ("mm_ConvNeXt", "frozen_fusion", "mm_MaxViT"):
            return model(image_input=img, metadata_input=meta)
        if kind in ("ConvNeXt", "MaxViT"):
            return model(input_data=img)
        return model(input_data=meta)


def maxvit_goldens(ref, ex):
    """MaxViT / mm_MaxViT: the reference's wrapper classes (resize, head surgery, metadata and
    fusion heads) around the stand-in backbone; the functional oracle must agree."""
    eimg = torch.from_numpy(ex["triplets"][[0, 1, 4, 5]])
    emeta = torch.from_numpy(ex["metadata"][[0, 1, 4, 5]])
    simg, smeta, _ = synthetic_batch(3, seed=2)
    out = {}
    for name, (kind, cfg) in MV_CONFIGS.items():
        sd = seeded_state_mv(kind, cfg, seed=3)
        model = getattr(ref, kind)(cfg).eval()
        model.load_state_dict(sd, strict=True)
        out[f"{name}/example4"] = call(kind, model, eimg, emeta).numpy()
        out[f"{name}/synthetic3"] = call(kind, model, simg, smeta).numpy()
        out[f"{name}/checksum"] = np.array(checksum(sd))
        with torch.no_grad():
            o = MO.forward(kind, sd, cfg, eimg, emeta)
        r = torch.from_numpy(out[f"{name}/example4"])
        err = (o - r).abs().max().item()
        scale = max(1.0, r.abs().max().item())
        print(f"{name}: reference-wrapper vs oracle max|dlogit| = {err:.2e} (max|logit| {scale:.1f})")
        assert err < 2e-5 * scale, name
    np.savez_compressed(os.path.join(HERE, "ref_logits_maxvit.npz"), **out)


def main():
    torch.manual_seed(0)
    ref = load_reference_architectures()
    if len(sys.argv) > 1 and sys.argv[1] == "maxvit":      # only (re)generate the MaxViT file
        maxvit_goldens(ref, dict(np.load(os.path.join(HERE, "example8.npz"))))
        return
    ex = example8()
    np.savez_compressed(os.path.join(HERE, "example8.npz"), **ex)
    maxvit_goldens(ref, ex)
    eimg, emeta = torch.from_numpy(ex["triplets"]), torch.from_numpy(ex["metadata"])
    simg, smeta, _ = synthetic_batch(6, seed=2)

    out = {}
    for name, (kind, cfg) in CONFIGS.items():
        sd = seeded_state(kind, cfg, seed=3)
        model = getattr(ref, kind)(cfg).eval()
        model.load_state_dict(sd, strict=True)
        out[f"{name}/example8"] = call(kind, model, eimg, emeta).numpy()
        out[f"{name}/synthetic6"] = call(kind, model, simg, smeta).numpy()
        out[f"{name}/checksum"] = np.array(checksum(sd))
        # the functional oracle must agree with the reference wrapper it restates
        with torch.no_grad():
            o = O.forward(kind, sd, cfg, eimg, emeta)
        r = torch.from_numpy(out[f"{name}/example8"])
        err = (o - r).abs().max().item()
        scale = max(1.0, r.abs().max().item())
        print(f"{name}: reference-wrapper vs oracle max|dlogit| = {err:.2e} (max|logit| {scale:.1f})")
        assert err < 2e-5 * scale, name
    out["synthetic6/img_checksum"] = np.array(simg.double().abs().sum().item())
    np.savez_compressed(os.path.join(HERE, "ref_logits.npz"), **out)

    seqs = {f"{w},{e}": [float(x) for x in O.lr_sequence(1e-4, e, w)] for w, e in [(0, 6), (2, 8)]}
    with open(os.path.join(HERE, "lr_sequences.json"), "w") as f:
        json.dump(seqs, f, indent=1)

    g = torch.Generator().manual_seed(11)
    p0 = torch.randn(257, generator=g)
    grads = torch.randn(3, 257, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-4, betas=(0.99, 0.99))
    traj = []
    for s in range(3):
        p.grad = grads[s].clone()
        opt.step()
        traj.append(p.detach().clone())
    z = torch.randn(64, generator=g) * 3
    y = (torch.rand(64, generator=g) < 0.4).float()
    zz = z.clone().requires_grad_(True)
    loss = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([2.5]))(zz, y)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "adamw_bce.npz"), p0=p0.numpy(), grads=grads.numpy(),
                        traj=torch.stack(traj).numpy(), z=z.numpy(), y=y.numpy(),
                        loss=np.array(loss.item()), dz=zz.grad.numpy(), pos_weight=np.array(2.5))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
