"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel name (mean per dispatch)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        key = (k, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""))
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for key, cs in acc.items():
    if "igemm_kernel" not in key[0] and "attn_self" not in key[0]:
        continue
    print(key)
    print("   " + "  ".join(f"{c}={v[0] / v[1]:.4g}" for c, v in sorted(cs.items())))
