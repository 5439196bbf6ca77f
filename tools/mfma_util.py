"""MFMA utilisation of the pointwise-conv kernels from SQ counters (one rocprofv3 --pmc pass, --kernel-trace only):

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
            SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
            -d gpurun_out/mfma -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 \
            --maxvit-steps 1 --maxvit-batch 256

  python tools/mfma_util.py gpurun_out/mfma profiles/r01_mfma_util.json

Per kernel symbol (sums over its dispatches):
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
                   SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs and counts 16 per v_mfma_f32_16x16x32_bf16
                   (= its 16384 FLOP at the 1024 FLOP/cycle/SIMD dense peak, so busy share == share of the 2.5 PFLOP/s
                   peak); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (155 us launch -> 3.00 M = 8 x 375 k
                   cycles at 2.4 GHz).  ROCm 7.2 ships no gfx950 derived-metric section (MI355X_MICROARCH.md,
                   "rocprofv3 PMC slots"), hence the explicit formula.
  mfma_flops     = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 FLOP  (one MOP = 512 FLOP; cross-checked against the
                   algorithmic count of bench.py where a kernel is all-GEMM)
  tflops         = mfma_flops / kernel time (kernel-trace timestamps of the same pass)
  frac_of_peak   = tflops / 2500
"""
import collections, csv, glob, json, os, re, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_mfma_util.json"
path = max(glob.glob(os.path.join(src, "*", "*_counter_collection.csv")), key=os.path.getmtime)

NSIMD = 4 * 256


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)I", k)
    if m:
        return m.group(1) + k[k.index(m.group(1)) + len(m.group(1)):][:40]
    return re.sub(r"\(.*", "", k)[:70]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(path)):
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0

rows = []
for k, c in acc.items():
    n = len(disp[k])
    t_ns = sum(disp[k].values())
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if busy == 0 and mops == 0:
        continue
    row = {"kernel": k, "dispatches": n, "time_ms": round(t_ns / 1e6, 3),
           "mfma_busy": round(busy / (NSIMD * gui / 8), 4) if gui else None,
           "mfma_gflop": round(mops * 512 / 1e9, 2),
           "tflops": round(mops * 512 / t_ns / 1e3, 1) if t_ns else None,
           "frac_of_2500": round(mops * 512 / t_ns / 1e3 / 2500, 4) if t_ns else None,
           "valu_active_of_wave_cycles": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3) if wc else None,
           "raw": {n_: v for n_, v in c.items()}}
    rows.append(row)
rows.sort(key=lambda r: -r["time_ms"])
json.dump({"source": __doc__.split("\n\n")[0] + " (see tools/mfma_util.py)", "file": os.path.basename(path),
           "kernels": rows}, open(dst, "w"), indent=1)
for r in rows[:25]:
    print(f'{r["kernel"][:60]:60s} n={r["dispatches"]:4d} {r["time_ms"]:8.3f} ms  busy {r["mfma_busy"]}  '
          f'{r["tflops"]} TF  valu {r["valu_active_of_wave_cycles"]}')
