"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): python tools/trace_by_grid.py DIR [substr]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.OrderedDict()
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        g = (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        a = acc.setdefault(g, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for g, (n, t) in acc.items():
    print(f"{g[0]:60s} grid=({g[1]},{g[2]},{g[3]}) n={n:4d} avg={t / n:8.1f} us")
