"""HBM traffic of the GEMM family (LDS-tiled / warp-specialised / pre-split igemm + row GEMM) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads:
doubled; both counters are in KiB).  Usage: python tools/pmc_traffic.py FETCH_DIR WRITE_DIR LAUNCHES_PER_STEP KEY OUT.json"""
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and any(k in r.get("Kernel_Name", "") for k in ("igemm_kernel", "igemm_ws_kernel", "igemm_ps_kernel", "igemm_pw_kernel", "rgemm_kernel")):
                tot += float(r["Counter_Value"])
                n += 1
    return tot, n


if __name__ == "__main__":
    fd, wd, per_step, key, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    f_kib, nf = collect(fd, "FETCH_SIZE")
    w_kib, nw = collect(wd, "WRITE_SIZE")
    fetch_b = 2.0 * f_kib * 1024 / max(nf, 1)       # per launch, gfx950 correction x2
    write_b = w_kib * 1024 / max(nw, 1)
    res = dict(fetch_bytes_per_launch=fetch_b, write_bytes_per_launch=write_b,
               hbm_bytes_per_launch=fetch_b + write_b, hbm_bytes_per_step=(fetch_b + write_b) * per_step,
               launches_counted=[nf, nw], correction="FETCH_SIZE x2 (gfx950), KiB -> bytes")
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[key] = res
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(res))
