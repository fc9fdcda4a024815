"""Summarise a rocprofv3 --kernel-trace --stats CSV directory into a compact text table."""
import csv
import glob
import os
import sys


def main(d, out=None):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append(r)
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}"]
    for r in rows[:40]:
        name = r["Name"]
        name = name if len(name) <= 90 else name[:87] + "..."
        lines.append(f"{name:90s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:10.3f} "
                     f"{float(r['AverageNs']) / 1e3:9.2f} {100 * float(r['TotalDurationNs']) / tot:6.2f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
