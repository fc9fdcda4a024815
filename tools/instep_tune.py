"""In-step plan selection.  Isolated launch timings mispredict what a plan costs inside the hipGraph step (the chip is
power-limited there and the operands arrive from another XCD): this tool lets the step decide.

    python tools/instep_tune.py variants TABLE.json CANDS.json OUTDIR [R]  # OUTDIR/v2.json .. vR.json: every shape on its
                                                                          # 2nd .. R-th isolated candidate (default 3)
    (on the GPU box: tools/layer_multi.sh OUTDIR/v2.json OUTDIR/v3.json -> per_key.json per table)
    python tools/instep_tune.py pick TABLE.json RUNDIR [--min-gain 0.02]  # RUNDIR/{t0,t1,...}/per_key.json -> TABLE.json updated
"""
import json
import os
import sys


def variants(table_path, cands_path, outdir, max_rank=3):
    table = json.load(open(table_path))
    cands = json.load(open(cands_path))
    os.makedirs(outdir, exist_ok=True)
    for rank in range(2, int(max_rank) + 1):
        t = dict(table)
        n = 0
        for key, cs in cands.items():
            alts = [c for c in cs if [c[0], c[1]] != table.get(key)]
            if len(alts) >= rank - 1:
                t[key] = [alts[rank - 2][0], alts[rank - 2][1]]
                n += 1
        json.dump(t, open(os.path.join(outdir, f"v{rank}.json"), "w"), indent=0, sort_keys=True)
        print(f"v{rank}.json: {n} shapes moved to their isolated rank-{rank} plan")


def pick(table_path, rundir, min_gain):
    table = json.load(open(table_path))
    runs = sorted(d for d in os.listdir(rundir) if os.path.exists(os.path.join(rundir, d, "per_key.json")))
    base = json.load(open(os.path.join(rundir, runs[0], "per_key.json")))
    saved = 0.0
    for r in runs[1:]:
        other = json.load(open(os.path.join(rundir, r, "per_key.json")))
        for key, e in other.items():
            b = base.get(key)
            if b is None or (e["cfg"], e["sk"]) == (b["cfg"], b["sk"]):
                continue
            if e["us"] < (1.0 - min_gain) * b["us"]:
                print(f"{key:40s} {b['cfg']:2d}/{b['sk']:<2d} {b['us']:8.1f} us -> {e['cfg']:2d}/{e['sk']:<2d} {e['us']:8.1f} us  (x{e['n']})")
                saved += b["us"] - e["us"]
                base[key] = e
                table[key] = [e["cfg"], e["sk"]]
    json.dump(table, open(table_path, "w"), indent=0, sort_keys=True)
    print(f"in-step time saved per step: {saved:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "variants":
        variants(*sys.argv[2:6])
    else:
        mg = float(sys.argv[sys.argv.index("--min-gain") + 1]) if "--min-gain" in sys.argv else 0.02
        pick(sys.argv[2], sys.argv[3], mg)
