"""Copy the judged summaries of tools/round_end.sh (gpurun_out/final/) into profiles/ under a round tag:
kernel stats, the two PMC passes, the bench line, and the corrected per-kernel HBM-side traffic
(bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024; calibration: tools/pmc_calib)."""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def newest(pattern):   # gpurun merges every call's files into gpurun_out/: take the latest run's
    return max(glob.glob(pattern), key=os.path.getmtime)


F = newest("gpurun_out/final/pmc_fetch/*/*_counter_collection.csv")
W = newest("gpurun_out/final/pmc_write/*/*_counter_collection.csv")


def agg(path, name):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            a[r["Kernel_Name"]][0] += 1
            a[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in a.items()}


def family(k):
    m = re.search(r"(?:::|N_1\d+?)([a-z][a-z_0-9]*_kernel)", k)   # mangled (N_1<len>name) or demangled (::name<...>) symbols
    return m.group(1) if m else k.split("(")[0][:40]


fa, wa = agg(F, "FETCH_SIZE"), agg(W, "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 "
                 "--warmup 2 --no-cpu-baseline --train-steps 0",
       "correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE reads 1/2: tools/pmc_calib, "
                     "0.500 on dword, dwordx4 and the stem gather; WRITE_SIZE exact)",
       "kernels": {}}
for k in fa:
    if "_GLOBAL__N_1" in k or "anonymous namespace" in k:
        n, f = fa[k]
        w = wa.get(k, (0, 0.0))[1]
        out["kernels"][k] = {"family": family(k), "launches": n, "fetch_bytes": round(2 * f * 1024),
                             "write_bytes": round(w * 1024), "traffic_bytes": round(2 * f * 1024 + w * 1024)}
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
shutil.copy(newest("gpurun_out/final/trace/*/*_kernel_stats.csv"), f"profiles/{tag}_kernel_stats_bench_bf16_b1024.csv")
shutil.copy(F, f"profiles/{tag}_pmc_fetch_size.csv")
shutil.copy(W, f"profiles/{tag}_pmc_write_size.csv")
try:   # training-step kernel stats and the MFMA-utilisation pass (tools/round_end.sh's last two runs)
    shutil.copy(newest("gpurun_out/final/train_trace/*/*_kernel_stats.csv"), f"profiles/{tag}_kernel_stats_train_bf16_b1024.csv")
    import subprocess
    subprocess.run([sys.executable, "tools/mfma_util.py", "gpurun_out/final/mfma", f"profiles/{tag}_mfma_util.json"],
                   check=True, stdout=subprocess.DEVNULL)
except ValueError:
    print("no train_trace / mfma pass under gpurun_out/final (older round_end.sh)")
if glob.glob("gpurun_out/final/pmc_fetch_train/*/*_counter_collection.csv"):   # HBM-side bytes of a training step
    import subprocess
    subprocess.run([sys.executable, "tools/train_traffic.py", "gpurun_out/final/pmc_fetch_train", "gpurun_out/final/pmc_write_train",
                    f"profiles/{tag}_pmc_traffic_train.json"], check=True)
if os.path.exists(f"profiles/{tag}_pmc_traffic_train.json") and os.path.exists(f"profiles/{tag}_kernel_stats_train_bf16_b1024.csv"):
    import subprocess
    subprocess.run([sys.executable, "tools/hbm_kernels.py", tag], check=True, stdout=subprocess.DEVNULL)
if glob.glob("gpurun_out/final/mv_fetch/*/*_counter_collection.csv"):   # mm_MaxViT forward: per-kernel statistics + traffic
    import subprocess
    shutil.copy(newest("gpurun_out/final/mv_trace/*/*_kernel_stats.csv"), f"profiles/{tag}_maxvit_kernel_stats.csv")
    subprocess.run([sys.executable, "tools/maxvit_traffic.py", "gpurun_out/final/mv_fetch", "gpurun_out/final/mv_write",
                    "gpurun_out/final/mv_trace", "4", f"profiles/{tag}_maxvit_traffic.json"], check=True)
if glob.glob("gpurun_out/final/mvt_trace/*/*_kernel_stats.csv"):   # mm_MaxViT training step (64 alerts): per-kernel statistics
    shutil.copy(newest("gpurun_out/final/mvt_trace/*/*_kernel_stats.csv"), f"profiles/{tag}_maxvit_train_kernel_stats.csv")
for src, dst in (("stamps_maxvit.log", "maxvit_part_stamps.txt"), ("nano.log", "nano_bench.txt"), ("train_ab.log", "train_ab.txt"), ("stamps_nano.log", "stage_stamps_nano.txt"),
                 ("stamps.log", "stage_stamps.txt"), ("stamps_train.log", "stage_stamps_train.txt"),
                 ("train_s2_ab.log", "train_stage2_forward_ab.txt"), ("train_timeline.txt", "train_timeline.txt"),
                 ("train_f32_ab.log", "train_f32_ab.txt")):   # tools/nano_bench.py; train step default vs
    if os.path.exists(f"gpurun_out/final/{src}"):                                        # one stream + three-launch LN/dw backward
        shutil.copy(f"gpurun_out/final/{src}", f"profiles/{tag}_{dst}")
_lines = [l for l in open("gpurun_out/final/bench_default.log") if l.startswith("{")]
open(f"profiles/{tag}_bench.json", "w").write(_lines[0])          # everything measured (bench_detail: true)
if len(_lines) > 1:
    open(f"profiles/{tag}_bench_line.json", "w").write(_lines[-1])   # the contract's line (< 7 KB)
for k, v in sorted(out["kernels"].items(), key=lambda x: -x[1]["traffic_bytes"] * x[1]["launches"])[:10]:
    print(f'{v["family"]:24s} launches {v["launches"]:4d}  fetch {v["fetch_bytes"]/1e6:8.2f} MB  write {v["write_bytes"]/1e6:8.2f} MB')
# every JSON summary bench.py quotes carries the digest of the kernel sources it was collected under (bench.py marks a
# summary from other sources "stale": true)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted(glob.glob("btsbot_amd/csrc/*.hip") + glob.glob("btsbot_amd/csrc/*.h")):
        hsh.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    sha = hsh.hexdigest()[:16]
    for name in ("pmc_traffic", "mfma_util", "depthwise_norm_hbm", "pmc_traffic_train"):
        path = f"profiles/{tag}_{name}.json"
        if os.path.exists(path):
            d = json.load(open(path))
            d["csrc_sha16"] = sha
            json.dump(d, open(path, "w"), indent=1)
    print("csrc_sha16", sha)
except Exception as e:   # noqa: BLE001
    print("could not stamp the summaries:", e)

