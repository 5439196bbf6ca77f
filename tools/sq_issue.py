"""Issue picture of the three stage kernels from SQ counters (one rocprofv3 --pmc pass, --kernel-trace only):

  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
            SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/sq -- python3 bench.py ...
  python tools/sq_issue.py gpurun_out/sq profiles/r04_sq_issue.json

Per kernel (sums over its dispatches): the shares of SQ_WAVE_CYCLES a wave spends parked (SQ_WAIT_ANY: s_waitcnt /
barrier), stalled at issue (SQ_WAIT_INST_ANY) and issuing (SQ_ACTIVE_INST_ANY; of which VALU), and the vector
instructions per wave (SQ_INSTS_VALU counts MFMAs too; SQ_INSTS_MFMA alone beside it)."""
import collections, csv, glob, json, os, re, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sq"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04_sq_issue.json"
path = max(glob.glob(os.path.join(src, "*", "*_counter_collection.csv")), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if not any(s in k for s in ("stage0b", "stage1b", "stage2p", "s3_fc", "head16")):
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    nd[k].add(r["Dispatch_Id"])
out = {"source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                 "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace (own pass)", "kernels": []}
for k, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    m = re.search(r"(stage0b|stage1b|stage2p|s3_fc1|s3_fc2|head16)_kernel", k)
    out["kernels"].append({
        "kernel": k[:90], "family": m.group(0) if m else k[:30], "dispatches": len(nd[k]),
        "parked_share_of_wave_cycles": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
        "issue_stall_share": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
        "issuing_share": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
        "valu_issuing_share": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
        "insts_valu_per_dispatch": round(c.get("SQ_INSTS_VALU", 0) / max(1, len(nd[k]))),
        "insts_mfma_per_dispatch": round(c.get("SQ_INSTS_MFMA", 0) / max(1, len(nd[k]))),
        "raw": {n: v for n, v in c.items()}})
json.dump(out, open(dst, "w"), indent=1)
for e in out["kernels"]:
    print(e["family"], e["dispatches"], "parked", e["parked_share_of_wave_cycles"], "issue-stall", e["issue_stall_share"],
          "issuing", e["issuing_share"], "(valu", e["valu_issuing_share"], ")", "valu insts/dispatch", e["insts_valu_per_dispatch"],
          "mfma", e["insts_mfma_per_dispatch"])
