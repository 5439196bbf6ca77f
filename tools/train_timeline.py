"""Developer tool: one training step's kernels in time order, both hardware queues side by side, from a rocprofv3
--kernel-trace of tools/train_bench.py.  usage: python tools/train_timeline.py <trace dir> [step index from the end]
Columns: start (us from the step's first kernel), duration, queue, gap since the previous kernel of the SAME queue, name."""
import csv, glob, re, sys
import os
f = max(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'), key=os.path.getmtime)   # (the newest trace: gpurun merges runs into one directory)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
ad = [i for i, e in enumerate(ev) if 'adamw' in e[2]]
# a step = from behind the previous step's last adamw launch to this step's last adamw launch (two adamw launches per step)
seg = ev[ad[-2 * back - 1] + 1: ad[-2 * back + 1] + 1]
t0 = seg[0][0]
last = {}
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return n[:64]
mainq = ev[ad[-1]][3]
tot = {}
for a, b, n, q in seg:
    gap = (a - last[q]) / 1e3 if q in last else 0.0
    last[q] = b
    mark = 'C' if q == mainq else '  s'
    print(f"{(a - t0) / 1e3:8.1f} {(b - a) / 1e3:7.1f} {mark:3s} gap {gap:6.1f}  {short(n)}")
busy = {}
for a, b, n, q in seg:
    busy[q] = busy.get(q, 0.0) + (b - a) / 1e3
chain = busy.get(mainq, 0.0)
side = sum(v for q, v in busy.items() if q != mainq)
print(f"step wall {(seg[-1][1] - t0) / 1e3:.1f} us; kernels: {len(seg)} launches, chain queue busy {chain:.1f} us, other queues {side:.1f} us, "
      f"both {chain + side:.1f} us")
