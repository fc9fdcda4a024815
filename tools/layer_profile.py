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


def dump(latent, batch, path, graph=False):
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
            from dsml_thesis_amd.engine import plan_key
            calls.append(dict(name=name, M=a.M, N=a.N, K=a.K, conv=a.a_mode, tf=a.a_tf, epi=a.epi, cfg=a.tile_cfg,
                              sk=a.splitk, key=plan_key(a, a.M), batch=max(1, a.batch), raw=int(a.raw_slabs), compute=int(a.compute)))
        else:
            calls.append(dict(name=name))
    side = [c[3] for c in getattr(run.pg, "side_calls", None) or []]       # launches on the forked stream (parallel branch)
    json.dump(dict(calls=calls, marker_steps=3, side=side), open(path, "w"))
    # marker: three more steps whose dispatches we will read from the END of the trace (--graph: hipGraph replays, so the
    # durations and the step span are those of the captured step the samplers actually run)
    if graph:
        from dsml_thesis_amd.engine import GraphedProgram
        g = GraphedProgram(run.eager_step)
        run.step = g.replay
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()


KERNEL_OF = {"ldmk_igemm": ("igemm_kernel", "igemm_ws_kernel", "igemm_ps_kernel", "igemm_psc_kernel", "igemm_pw_kernel", "rgemm_kernel", "sgemm_kernel"), "ldmk_post": ("post_",),
             "ldmk_gn_apply_ps_h2": ("gn_apply_ps_h2",),
             "ldmk_attn_self_small": ("attn_small",), "ldmk_conv3x3_out_small": ("conv3x3_out_small",), "ldmk_gn_finalize": ("gn_finalize",), "ldmk_gn_apply": ("gn_apply",),
             "ldmk_gn_partial": ("gn_partial",), "ldmk_ln_stats": ("ln_stats",), "ldmk_ln_stats_guard": ("ln_stats",), "ldmk_ln_stats_split": ("ln_stats",), "ldmk_ln_stats_ps": ("ln_stats_ps",), "ldmk_ln_stats_ps_h2": ("ln_stats_ps",), "ldmk_pack_ps": ("pack_ps",), "ldmk_attn_self": ("attn_self",), "ldmk_attn_self_x3": ("attn_x3",), "ldmk_attn_self_x3p": ("attn_kv_split",), "ldmk_attn_self_x3p_ps": ("attn_kv_split",), "ldmk_attn_self_h2": ("attn_kv_split_h2",), "ldmk_attn_self_h2_ps": ("attn_kv_split_h2",), "ldmk_attn_self_h2_tiles": ("attn_h2_fwd",),
             "ldmk_attn_cross": ("attn_cross",), "ldmk_dense_small": ("dense_small",),
             "ldmk_timestep_embedding": ("timestep_embedding",), "ldmk_conv3x3_in": ("conv3x3_in",),
             "ldmk_conv3x3_out": ("conv3x3_out",), "ldmk_winograd_input": ("wino_input",), "ldmk_winograd_input_ps": ("wino_input_ps",), "ldmk_upconv_gather_ps": ("upconv_gather_ps",), "ldmk_winograd_input_ps_h2": ("wino_input_ps",), "ldmk_upconv_gather_ps_h2": ("upconv_gather_ps",),
             "ldmk_winograd_output": ("wino_output",), "ldmk_upconv_gather": ("upconv_gather",),
             "ldmk_upconv_scatter": ("upconv_scatter",)}
PEAK_F32_MFMA = 157.3
PEAK_BF16_MFMA = 2516.6


def join(d):
    """Walk the trace and the launch program side by side, matching kernel NAMES (a split-K GEMM is its main kernel
    plus the reduce kernel that follows it); any mismatch is an error, and so is a GEMM row above the matrix peak."""
    prog = json.load(open(os.path.join(d, "prog.json")))
    calls = prog["calls"]
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(tr)) if "ldmk::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = sum(2 if (c["name"] in ("ldmk_post", "ldmk_attn_self_x3p", "ldmk_attn_self_x3p_ps", "ldmk_attn_self_h2", "ldmk_attn_self_h2_ps") or (c["name"] == "ldmk_igemm" and c.get("sk", 1) > 1 and not 7 <= c.get("cfg", 0) <= 12)) else 1
              for c in calls) + 1
    # the last step of the trace: find it by walking back from the end to the step's first kernel (timestep_embedding)
    starts = [i for i, r in enumerate(rows) if "timestep_embedding" in r["Kernel_Name"]]
    last = rows[starts[-1]:]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    # launches of the parallel branch (timestep embedding + emb_layers on the forked stream) interleave with the main chain
    # in the trace: take them out of the walk and list them on their own
    side_kernels = tuple(k for name in prog.get("side", []) for k in KERNEL_OF.get(name, ()))
    side_rows = [r for r in last if side_kernels and any(k in r["Kernel_Name"] for k in side_kernels)]
    if side_rows:
        last = [r for r in last if r not in side_rows]
    i = 0
    out = []
    for c in calls:
        want = KERNEL_OF.get(c["name"])
        r = last[i]
        assert want is None or any(w in r["Kernel_Name"] for w in want), (c, r["Kernel_Name"])
        d_ = dur(r)
        i += 1
        if c["name"] in ("ldmk_attn_self_x3p", "ldmk_attn_self_x3p_ps", "ldmk_attn_self_h2", "ldmk_attn_self_h2_ps"):          # K / V pre-pass + attention kernel
            assert ("attn_h2_fwd" if c["name"].startswith("ldmk_attn_self_h2") else "attn_x3p") in last[i]["Kernel_Name"], last[i]["Kernel_Name"]
            c["prepass_us"] = d_
            d_ += dur(last[i])
            i += 1
        if c["name"] == "ldmk_post" and "post_gnstat" in r["Kernel_Name"]:       # row-tiled GroupNorm: statistics + apply launches
            assert "post_gnapply" in last[i]["Kernel_Name"], last[i]["Kernel_Name"]
            d_ += dur(last[i])
            i += 1
        if (c["name"] == "ldmk_igemm" and any(k in r["Kernel_Name"] for k in ("igemm_kernel", "igemm_ws_kernel", "igemm_ps_kernel", "igemm_psc_kernel", "igemm_pw_kernel", "sgemm_kernel"))
                and c.get("sk", 1) > 1 and not c.get("raw")):
            assert "igemm_reduce" in last[i]["Kernel_Name"], (c, last[i]["Kernel_Name"])
            c["main_us"] = d_
            d_ += dur(last[i])
            i += 1
        out.append((c, d_))
    assert per >= i, (per, i)
    tot = sum(d_ for _, d_ in out)
    span = (int(last[i - 1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
    print(f"step total (sum of kernel durations) {tot / 1e3:.3f} ms over {len(out)} calls, {i} kernel launches; "
          f"first start to last end {span:.3f} ms")
    # machine-readable twin for tools/instep_tune.py: per plan key the plan that ran and its total time in the step
    per_key = {}
    for c, d_ in out:
        if c["name"] == "ldmk_igemm" and "key" in c:
            e = per_key.setdefault(c["key"], dict(cfg=c["cfg"], sk=c.get("sk", 1), us=0.0, n=0))
            e["us"] += d_
            e["n"] += 1
    json.dump(per_key, open(os.path.join(d, "per_key.json"), "w"), indent=0, sort_keys=True)
    agg = {}
    for c, d_ in out:
        if c["name"] == "ldmk_igemm":
            key = f"igemm M={c['M']:6d} N={c['N']:5d} K={c['K']:6d} conv={c['conv']} tf={c['tf']} epi={c['epi']} cfg={c['cfg']} sk={c.get('sk', 1)}"
            if c.get("batch", 1) > 1:
                key += f" x{c['batch']} ({'Winograd' if c['batch'] == 16 else 'upsample phases'})"
            if c.get("compute", 0) == 2:
                key += " bf16x3"
            elif c.get("compute", 0) == 3:
                key += " f16x2"
        else:
            key = c["name"]
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += d_
        a[3] += c.get("main_us", d_)
        if c["name"] == "ldmk_igemm":
            a[2] += 2.0 * c["M"] * c["N"] * c["K"] * c.get("batch", 1)
    if side_rows:
        print(f"parallel branch (forked stream, not on the critical path): {len(side_rows)} launches, "
              f"{sum(dur(r) for r in side_rows):.1f} us")
    print(f"{'call':75s} {'n':>3s} {'us_total':>10s} {'pct':>6s} {'TFLOP/s':>8s}")
    fam_t = fam_f = 0.0
    for key, (n, d_, fl, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tfs = fl / (d_ * 1e-6) / 1e12 if fl else 0.0
        # bf16x3 rows: six bf16 MFMAs per fp32-equivalent product -> ceiling = bf16 peak / 6
        peak = PEAK_BF16_MFMA / 6.0 if key.endswith("bf16x3") else PEAK_BF16_MFMA / 3.0 if key.endswith("f16x2") else PEAK_F32_MFMA
        assert tfs <= peak, f"{key}: {tfs:.1f} TFLOP/s exceeds the matrix peak of its arithmetic ({peak:.1f}) -- the join is wrong"
        fam_t += d_ if fl else 0.0
        fam_f += fl
        print(f"{key:75s} {n:3d} {d_:10.1f} {100 * d_ / tot:6.2f} {tfs:8.1f}  main {mn / n:6.1f} us/call" if fl else f"{key:75s} {n:3d} {d_:10.1f} {100 * d_ / tot:6.2f}")
    print(f"GEMM family (LDS-tiled igemm + row GEMM): {fam_f * 1e-9:.1f} GFLOP (fp32-equivalent 2MNK) in {fam_t / 1e3:.3f} ms = "
          f"{fam_f / (fam_t * 1e-6) / 1e12:.1f} TFLOP/s = {fam_f / (fam_t * 1e-6) / 1e12 / PEAK_F32_MFMA:.3f} of the f32 matrix peak "
          f"(rows marked bf16x3 / f16x2 execute 6 / 3 16-bit MFMA FLOPs per counted FLOP: their ceilings are {PEAK_BF16_MFMA / 6:.1f} / {PEAK_BF16_MFMA / 3:.1f})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dump")
    ap.add_argument("--join")
    ap.add_argument("--graph", action="store_true", help="profile hipGraph replays of the step instead of eager launches")
    a = ap.parse_args()
    if a.dump:
        dump(a.latent, a.batch, a.dump, a.graph)
    else:
        join(a.join)
