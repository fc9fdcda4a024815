"""Per-launch view of one UNet step: joins the launch program's igemm list (M, N, K, tile) with the
per-dispatch durations of a rocprofv3 --kernel-trace run of this very script.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/layer_profile.py --latent 64 --dump OUT/prog.json
    python tools/layer_profile.py --join OUT
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dump(latent, batch, path):
    import torch
    from bench import StepRunner, build_model
    model, ucfg = build_model(latent, torch.device("cuda", 0))
    run = StepRunner(model, ucfg, batch, graph=False)
    for _ in range(2):
        run.step()
    torch.cuda.synchronize()
    calls = []
    for fn, args, keep, name in run.pg.calls:
        if name == "ldmk_igemm":
            a = keep
            calls.append(dict(name=name, M=a.M, N=a.N, K=a.K, conv=a.a_mode, tf=a.a_tf, epi=a.epi, cfg=a.tile_cfg,
                              sk=a.splitk))
        else:
            calls.append(dict(name=name))
    json.dump(dict(calls=calls, marker_steps=3), open(path, "w"))
    # marker: three more steps whose dispatches we will read from the END of the trace
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()


def join(d):
    prog = json.load(open(os.path.join(d, "prog.json")))
    calls = prog["calls"]
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(tr)) if "ldmk::" in r["Kernel_Name"]]
    # expected ldmk dispatches per step
    per = 0
    for c in calls:
        per += 2 if (c["name"] == "ldmk_gn_coef" or c.get("sk", 1) > 1) else 1
    per += 2  # ddim step + advance
    last = rows[-per:]
    i = 0
    out = []
    for c in calls:
        k = 2 if (c["name"] == "ldmk_gn_coef" or c.get("sk", 1) > 1) else 1
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last[i:i + k]) / 1e3
        i += k
        out.append((c, dur))
    tot = sum(d_ for _, d_ in out)
    print(f"step total (sum of kernel durations) {tot / 1e3:.3f} ms over {len(out)} calls")
    agg = {}
    lines = []
    for c, d_ in out:
        if c["name"] == "ldmk_igemm":
            fl = 2.0 * c["M"] * c["N"] * c["K"]
            tf = fl / (d_ * 1e-6) / 1e12
            key = f"igemm M={c['M']:6d} N={c['N']:5d} K={c['K']:6d} conv={c['conv']} tf={c['tf']} epi={c['epi']} cfg={c['cfg']} sk={c.get('sk', 1)}"
            lines.append((key, d_, tf))
        else:
            key = c["name"]
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += d_
        if c["name"] == "ldmk_igemm":
            a[2] += 2.0 * c["M"] * c["N"] * c["K"]
    print(f"{'call':75s} {'n':>3s} {'us_total':>10s} {'pct':>6s} {'TFLOP/s':>8s}")
    for key, (n, d_, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tfs = f"{fl / (d_ * 1e-6) / 1e12:8.1f}" if fl else "        "
        print(f"{key:75s} {n:3d} {d_:10.1f} {100 * d_ / tot:6.2f} {tfs}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dump")
    ap.add_argument("--join")
    a = ap.parse_args()
    if a.dump:
        dump(a.latent, a.batch, a.dump)
    else:
        join(a.join)
