"""Print the top kernels of a rocprofv3 --kernel-trace --stats run.  usage: kstats.py <dir> [steps] [top]"""
import csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
f = sorted(glob.glob(d + "/*/*_kernel_stats.csv") + glob.glob(d + "/*_kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: {tot / 1e6 / steps:.3f} ms of kernel time per step over {steps:.0f} steps")
for r in rows[:top]:
    print(r["Name"][:86].ljust(86), r["Calls"].rjust(6), "%9.1f us" % (float(r["AverageNs"]) / 1e3),
          "%8.3f ms/step" % (float(r["TotalDurationNs"]) / 1e6 / steps))
