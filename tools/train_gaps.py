"""Developer tool: where a training step's wall time goes, from a rocprofv3 --kernel-trace of tools/train_bench.py.
Per step: wall, union-busy and idle time of the GPU, busy time per HW queue, idle time by the kernel that ends each gap,
and the chain's (queue of the optimiser kernel) kernels by total time."""
import collections, csv, glob, sys
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_train') + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
ad = [i for i, e in enumerate(ev) if 'adamw' in e[2]]
NS = 10
seg = ev[ad[-2 * NS - 1] + 1:ad[-1] + 1]
t0, t1 = seg[0][0], seg[-1][1]
busy, gaps = 0, collections.Counter()
cs, ce = seg[0][0], seg[0][1]
for a, b, n, q in seg[1:]:
    if a > ce:
        busy += ce - cs
        gaps[n[:60]] += a - ce
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
print(f"per step: wall {(t1 - t0) / NS / 1e3:.1f} us, busy {busy / NS / 1e3:.1f}, idle {(t1 - t0 - busy) / NS / 1e3:.1f}")
qs = collections.Counter()
for a, b, n, q in seg:
    qs[q] += b - a
print("busy per queue (us/step):", {q: round(v / NS / 1e3, 1) for q, v in qs.items()})
print("idle before (us/step):", [(n, round(v / NS / 1e3, 1)) for n, v in gaps.most_common(10)])
mainq = ev[ad[-1]][3]
kt, kc = collections.Counter(), collections.Counter()
for a, b, n, q in seg:
    if q == mainq:
        kt[n[:70]] += b - a
        kc[n[:70]] += 1
print(f"chain queue {mainq}:")
for n, v in kt.most_common(30):
    print(f"  {n:70s} {kc[n] / NS:5.1f}/step {v / NS / 1e3:8.1f} us/step")
