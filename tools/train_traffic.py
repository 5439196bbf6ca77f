"""HBM-side bytes of ONE training step from two rocprofv3 passes over tools/train_bench.py (tools/round_end.sh profile:
gpurun_out/final/pmc_fetch_train, pmc_write_train), with the guide's gfx950 correction (bytes = 2 * FETCH_SIZE * 1024 +
WRITE_SIZE * 1024; MI355X_MICROARCH.md, HBM).  Steps are counted by the launches of bce_kernel (one per step).
usage: train_traffic.py <fetch dir> <write dir> <out.json>"""
import collections, csv, glob, json, re, sys

fdir, wdir, out = sys.argv[1:4]


def load(d, name):
    import os
    f = max(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)   # gpurun merges every call's files
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            a[r["Kernel_Name"]][0] += 1
            a[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return a


def family(k):
    k = k.replace("(anonymous namespace)::", "")
    m = re.search(r"\d+([a-z_0-9]+_kernel)", k)
    return m.group(1) if m else re.sub(r"^void ", "", k).split("(")[0].split("<")[0][:48]


fa, wa = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
steps_f = sum(n for k, (n, _) in fa.items() if "bce_kernel" in k)
steps_w = sum(n for k, (n, _) in wa.items() if "bce_kernel" in k)
fam = collections.defaultdict(lambda: {"launches_per_step": 0.0, "fetch_bytes": 0.0, "write_bytes": 0.0})
for k, (n, v) in fa.items():
    e = fam[family(k)]
    e["launches_per_step"] += n / steps_f
    e["fetch_bytes"] += 2 * v * 1024 / steps_f
for k, (n, v) in wa.items():
    fam[family(k)]["write_bytes"] += v * 1024 / steps_w
for e in fam.values():
    e["traffic_bytes"] = round(e["fetch_bytes"] + e["write_bytes"])
    e["fetch_bytes"], e["write_bytes"] = round(e["fetch_bytes"]), round(e["write_bytes"])
    e["launches_per_step"] = round(e["launches_per_step"], 2)
tot_f = sum(e["fetch_bytes"] for e in fam.values())
tot_w = sum(e["write_bytes"] for e in fam.values())
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (separate passes) -- python3 tools/train_bench.py 1024 bf16 5",
       "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reads 1/2 of a wide streaming read)",
       "steps_counted": [steps_f, steps_w], "per_step": {"fetch_bytes": tot_f, "write_bytes": tot_w, "traffic_bytes": tot_f + tot_w},
       "families": dict(sorted(fam.items(), key=lambda kv: -kv[1]["traffic_bytes"]))}
json.dump(res, open(out, "w"), indent=1)
print(f"per step: fetch {tot_f/1e9:.2f} GB, write {tot_w/1e9:.2f} GB, total {(tot_f+tot_w)/1e9:.2f} GB over {steps_f} steps")
for k, e in list(res["families"].items())[:12]:
    print(f'{k:40s} {e["launches_per_step"]:6.1f} launches  {e["traffic_bytes"]/1e6:9.1f} MB')
