"""Per-kernel HBM-side bytes of ONE mm_MaxViT forward from two rocprofv3 PMC passes over tools/mv_bench.py (separate
FETCH_SIZE / WRITE_SIZE runs; bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024, the gfx950 correction of
MI355X_MICROARCH.md / tools/pmc_calib), next to the kernel trace's durations of the same command.
usage: maxvit_traffic.py <fetch dir> <write dir> <stats dir> <forwards per run> <out.json>"""
import collections, csv, glob, json, os, re, sys

fdir, wdir, sdir, nfw, out = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]


def newest(pat):
    return max(glob.glob(pat), key=os.path.getmtime)


def load(d, name):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(newest(f"{d}/*/*_counter_collection.csv"))):
        if r["Counter_Name"] == name:
            a[r["Kernel_Name"]][0] += 1
            a[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return a


def family(k):
    # mangled (_ZN12_GLOBAL__N_1<len><name>...) or demangled (...::<name><...>(...)) symbols; names may hold digits (mv_part64_kernel)
    m = re.search(r"_GLOBAL__N_1(\d+)", k)
    if m:
        n = int(m.group(1))
        for cut in (2, 1, 3):   # the length prefix is 1-3 digits: take the split whose length matches a *_kernel name
            pre = m.group(1)[:cut]
            name = k[m.start(1) + cut:m.start(1) + cut + int(pre)] if pre.isdigit() else ""
            if name.endswith("_kernel") and len(name) == int(pre):
                return name
    k = k.replace("(anonymous namespace)::", "")
    return re.sub(r"^void ", "", k).split("(")[0].split("<")[0][:48]


fa, wa = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
fam = collections.defaultdict(lambda: {"launches": 0.0, "fetch_bytes": 0.0, "write_bytes": 0.0, "ms": 0.0})
for k, (n, v) in fa.items():
    e = fam[family(k)]
    e["launches"] += n / nfw
    e["fetch_bytes"] += 2 * v * 1024 / nfw
for k, (n, v) in wa.items():
    fam[family(k)]["write_bytes"] += v * 1024 / nfw
for r in csv.DictReader(open(newest(f"{sdir}/*/*_kernel_stats.csv"))):
    fam[family(r["Name"])]["ms"] += float(r["TotalDurationNs"]) / 1e6 / nfw
res = {}
for k, e in fam.items():
    if e["ms"] <= 0 or not (k.startswith("mv_") or "gemm" in k or "fused_mlp" in k):
        continue
    tb = e["fetch_bytes"] + e["write_bytes"]
    res[k] = {"launches_per_forward": round(e["launches"], 1), "ms_per_forward": round(e["ms"], 3),
              "traffic_mb": round(tb / 1e6, 1), "hbm_gbs": round(tb / (e["ms"] * 1e-3) / 1e9, 1) if e["ms"] > 0 else None}
tot_ms = sum(v["ms_per_forward"] for v in res.values())
tot_mb = sum(v["traffic_mb"] for v in res.values())
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --kernel-trace --stats (separate passes) -- python3 tools/mv_bench.py 1024 bf16 3",
       "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024", "forwards_per_run": nfw,
       "per_forward": {"kernel_ms": round(tot_ms, 2), "traffic_gb": round(tot_mb / 1e3, 2),
                       "mean_hbm_gbs": round(tot_mb / 1e3 / (tot_ms * 1e-3), 1) if tot_ms > 0 else None},
       "families": dict(sorted(res.items(), key=lambda kv: -kv[1]["ms_per_forward"]))}
json.dump(doc, open(out, "w"), indent=1)
print(f"per forward of 1024 alerts: {tot_ms:.2f} ms of kernels, {tot_mb / 1e3:.2f} GB HBM-side")
for k, v in list(doc["families"].items())[:14]:
    print(f'{k:34s} {v["launches_per_forward"]:5.1f} launches {v["ms_per_forward"]:7.3f} ms {v["traffic_mb"]:9.1f} MB {v["hbm_gbs"]:8.1f} GB/s')
