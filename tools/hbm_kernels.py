"""Achieved HBM rate of the depthwise / normalisation kernels of a training step (the kernels BASELINE.json's north star
asks an HBM figure for): HBM-side bytes per launch from the PMC passes (profiles/<tag>_pmc_traffic_train.json, corrected
as MI355X_MICROARCH.md prescribes) over the average launch duration of the kernel trace of the same workload
(profiles/<tag>_kernel_stats_train_bf16_b1024.csv), against the 8 TB/s peak.  The backward kernels share the chip with
the side stream's filter-gradient GEMMs, so their durations are what they take in the step, not in isolation.
usage: hbm_kernels.py <tag>  ->  profiles/<tag>_depthwise_norm_hbm.json"""
import collections, csv, json, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
HBM_PEAK_GBS = 8000.0
FAMILIES = ("dwconv_ln_kernel", "dw3_ln_kernel", "dw1_ln_kernel", "dwln_bwd_kernel", "ln_dw1_bwd_kernel",
            "ln_bwd_narrow_kernel", "ln_bwd_kernel", "ln_patch_kernel", "stem16_kernel", "adamw_kernel")


def family(k):
    k = k.replace("(anonymous namespace)::", "")
    m = re.search(r"\d+([a-z_0-9]+_kernel)", k)
    return m.group(1) if m else re.sub(r"^void ", "", k).split("(")[0].split("<")[0][:48]


traffic = json.load(open(f"profiles/{tag}_pmc_traffic_train.json"))["families"]
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats_train_bf16_b1024.csv")):
    f = family(r["Name"])
    dur[f][0] += int(r["Calls"])
    dur[f][1] += float(r["TotalDurationNs"])
out = {"source": f"profiles/{tag}_pmc_traffic_train.json (bytes) / profiles/{tag}_kernel_stats_train_bf16_b1024.csv (durations)",
       "peak_gbs": HBM_PEAK_GBS, "kernels": {}}
for f in FAMILIES:
    if f not in traffic or dur[f][0] == 0:
        continue
    t = traffic[f]
    per_launch = t["traffic_bytes"] / t["launches_per_step"]
    avg_us = dur[f][1] / dur[f][0] / 1e3
    gbs = per_launch / avg_us / 1e3
    out["kernels"][f] = {"launches_per_step": t["launches_per_step"], "hbm_bytes_per_launch": round(per_launch),
                         "avg_launch_us": round(avg_us, 2), "achieved_gbs": round(gbs, 1),
                         "frac_of_peak": round(gbs / HBM_PEAK_GBS, 3)}
json.dump(out, open(f"profiles/{tag}_depthwise_norm_hbm.json", "w"), indent=1)
for f, e in out["kernels"].items():
    print(f'{f:24s} {e["launches_per_step"]:5.1f}/step  {e["hbm_bytes_per_launch"]/1e6:8.1f} MB  {e["avg_launch_us"]:7.1f} us  '
          f'{e["achieved_gbs"]:7.0f} GB/s  {e["frac_of_peak"]:.3f}')
