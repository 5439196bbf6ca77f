"""Summarise a rocprofv3 kernel_trace.csv: device time per (kernel, grid) in dispatch order of one step."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0][:70]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", ""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0, 10**18])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d)
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{a[1]/tot*100:5.1f}%  n={a[0]:5d}  avg={a[1]/a[0]/1e3:8.2f}us  min={a[2]/1e3:8.2f}us  grid=({k[1]},{k[2]}) wg={k[3]}  {k[0]}")
