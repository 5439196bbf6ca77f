import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
if len(sys.argv) > 1:
    from helpers import CONFIGS, seeded_state, build_model, run_model
    from btsbot_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    kind, cfg = CONFIGS["mm_nano_ls"]
    out = {}
    for B in (1, 4, 5, 6, 2053, 4099):
        img, meta, _ = synthetic_batch(B, seed=5)
        for prec in ("bf16", "f16"):
            m = build_model(kind, cfg, seeded_state(kind, cfg, seed=3), dev, prec)
            o = run_model(kind, m, img.to(dev), meta.to(dev)).float().cpu()
            out[f"{B}_{prec}"] = o.flatten().tolist()
    json.dump(out, open(sys.argv[1], "w"))
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/a.json"], env=env)
    env["BTSBOT_AMD_NO_STAGE2"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/b.json"], env=env)
    a, b = json.load(open("/tmp/a.json")), json.load(open("/tmp/b.json"))
    for k in a:
        x, y = torch.tensor(a[k]), torch.tensor(b[k])
        print(k, "max|dlogit| fused vs per-op", float((x - y).abs().max()), "scale", float(y.abs().max()), "finite", bool(torch.isfinite(x).all()))
