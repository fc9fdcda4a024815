"""Per-kernel means of the counters in a rocprofv3 --pmc run stored as a rocpd SQLite database (ROCm 7.2 default).
    python tools/pmc_db.py <dir-or-db> [kernel-name-substring]"""
import collections
import glob
import os
import sqlite3
import sys


def tables(con, prefix):
    return [r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith(prefix)]


def summarize(path, match=""):
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for db in dbs:
        con = sqlite3.connect(db)
        ev, pmc = tables(con, "rocpd_pmc_event")[0], tables(con, "rocpd_info_pmc")[0]
        kd, ks = tables(con, "rocpd_kernel_dispatch")[0], tables(con, "rocpd_info_kernel_symbol")[0]
        q = (f"select s.kernel_name, d.grid_size_x, d.dispatch_id, p.name, sum(e.value), d.end - d.start from {ev} e "
             f"join {pmc} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
             f"group by d.dispatch_id, p.name")
        seen = set()
        for name, grid, did, cname, val, ns in con.execute(q):
            if match not in name:
                continue
            key = (name[:90], grid)
            a = acc[key][cname]
            a[0] += val
            a[1] += 1
            if did not in seen:
                seen.add(did)
                dur[key][0] += ns
                dur[key][1] += 1
    return acc, dur


if __name__ == "__main__":
    acc, dur = summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for key, cs in acc.items():
        d = dur[key]
        print(key, f"dispatches={d[1]} avg_us={d[0] / max(d[1], 1) / 1e3:.1f}")
        print("   " + "  ".join(f"{c}={v[0] / v[1]:.5g}" for c, v in sorted(cs.items())))
