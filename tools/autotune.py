"""Offline tuning of the igemm (tile shape, split-K) plan per problem shape.

    python tools/autotune.py --case 64:16 [--case 32:16 ...] [--out gpurun_out/plans_f32.json]
    (flat files; the sweep starts from the matching section of dsml_thesis_amd/igemm_plans.json and
     `python tools/merge_plans.py --merge <section> <flat>` puts a result back)

For every distinct igemm problem of the UNet launch program (M, N, K, conv/rows, prologue, epilogue) every legal
(tile_cfg, splitk) pair is timed on the real buffers (HIP events, median of 5) and the fastest is recorded.
The table is a static, committed artefact: `engine.Program.igemm` looks a shape up there first and falls back to
the C++ heuristic (`ldmk_igemm_plan`) for shapes it has never seen -- plans stay deterministic, so results remain
bitwise reproducible across runs and ranks.
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ROWS_RETUNE = False
SLAB_RETUNE = False
FORCE = False
X3_TABLE = None
H2_MODE = False       # --h2: the --x3 sweep in the F16X2 arithmetic
H2_FLAG = None
CANDS = {}
CFG_WK = {1: 1, 2: 2, 3: 4, 4: 2, 5: 1, 6: 2}          # K slices of 32 staged per iteration
EVEN_TN = {1, 2, 3}
SKS = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64]


def time_call(lib, a, st, reps=7, touch=()):
    """Median launch time.  `touch`: the A operands as tensors -- rewritten in place by an elementwise kernel before every
    timed launch, as the producing layer does inside the step.  Without it the repeated launch finds its A rows in the L2
    of the very XCD that read them a moment ago and A-heavy tiles look ~20 % better than they run in the hipGraph
    (per-XCD L2s: what a producer wrote on another XCD comes from the fabric)."""
    ts = []
    for _ in range(2):
        if lib.ldmk_igemm(C.byref(a), st) != 0:
            return None
    torch.cuda.synchronize()
    for _ in range(reps):
        for t in touch:
            t.mul_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.ldmk_igemm(C.byref(a), st)
        e1.record()
        torch.cuda.synchronize()
        if rc != 0:
            return None
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def tune(kind, latent, batch, table):
    from bench import StepRunner, build_model
    model, ucfg = build_model(latent, torch.device("cuda", 0))
    if kind == "train":
        # the eager training step: record every GEMM it issues (forward, data gradients), then sweep the distinct shapes
        from dsml_thesis_amd import train as TR
        tr = TR.UNetTrainer(model.model.diffusion_model)
        dev = torch.device("cuda", 0)
        g = torch.Generator().manual_seed(0)
        x0 = torch.randn(batch, ucfg["in_channels"], latent, latent, generator=g).to(dev)
        noise = torch.randn(batch, ucfg["out_channels"], latent, latent, generator=g).to(dev)
        ctx = torch.randn(batch, 1, ucfg["context_dim"], generator=g).to(dev)
        t = torch.randint(0, 1000, (batch,), generator=g).to(dev)
        TR.RECORD = []
        # keep every tensor of the step alive so the recorded pointers stay valid during the sweep
        tr.p_losses(x0, ctx, t, noise, model.sqrt_alphas_cumprod, model.sqrt_one_minus_alphas_cumprod)
        rec, TR.RECORD = TR.RECORD, None
        torch.cuda.synchronize()

        class _PG:
            pass
        pg = _PG()
        pg.lib = __import__("dsml_thesis_amd.lib", fromlist=["load"]).load()
        big = torch.empty(512 * 1024 * 1024 // 4, device=dev)      # recorded pointers may be recycled: sweep on scratch operands
        pg.calls = [(None, None, a, "ldmk_igemm") for a in rec]
        for a in rec:
            a.a0 = big.data_ptr()
            a.a1 = big.data_ptr() if a.c1 else 0
            a.w = big.data_ptr()
            a.bias = a.batch_vec = a.residual = a.tf_coef = a.row_stats = a.ln_gamma = a.ln_beta = 0
            a.a_tf = 0
        tune_program(pg, table)
        return
    if kind == "unet":
        run = StepRunner(model, ucfg, batch, graph=False)
        run.step()
        pg = run.pg
    else:
        fs = model.first_stage_model
        if kind == "dec":
            fs.decode(torch.randn(batch, ucfg["in_channels"], latent, latent, device="cuda"))
            pg = fs._program("dec", batch, latent, latent, True)
        else:
            fs.encode(torch.randn(batch, 3, latent * 4, latent * 4, device="cuda"))
            pg = fs._program("enc", batch, latent * 4, latent * 4)
    torch.cuda.synchronize()
    tune_program(pg, table)


def tune_x3(pg, xtable):
    """--x3: for every GEMM of the program whose weight has bf16x3 split images, the best LDMK_COMPUTE_BF16X3 plan against the
    plan the program runs today (f32: LDS-tiled, row GEMM or slab GEMM, as recorded); shapes where the split arithmetic wins
    by > 3 % go to the bf16x3 section of the plan file, which engine.Program.plan() consults first."""
    from dsml_thesis_amd import lib as L, ops
    from dsml_thesis_amd.engine import plan_key
    lib = pg.lib
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(16 * 1024 * 1024 * 16, device="cuda")
    seen = set()
    for fn, args, a, name in pg.calls:
        if name != "ldmk_igemm":
            continue
        key = plan_key(a, a.M)
        if key in seen or a.b_trans or a.raw_slabs or a.compute != L.COMPUTE_F32 or (ops.split_h2_of(a.w) if H2_MODE else ops.split_of(a.w)) is None:
            continue
        seen.add(key)
        saved = (a.tile_cfg, a.splitk, a.splitk_ws, a.splitk_ws_elems, a.out, a.stats_out, a.residual, a.compute, a.w_split,
                 a.w_split_ld, a.w_split_bstride)
        saved_h = (a.w_scale_exp, a.range_flag, a.a_split, a.a_split_ld)
        saved_a = (a.a0, a.a1)
        nb = max(1, a.batch)
        samples = -(-a.M // a.rows_per_sample)
        rows = samples * a.in_h * a.in_w if a.a_mode == 1 else a.M
        span = (nb - 1) * a.a_bstride if nb > 1 else 0
        sa0 = torch.randn(rows * a.c0 + span, device="cuda")
        sa1 = torch.randn(rows * a.c1 + span, device="cuda") if a.c1 else None
        a.a0 = sa0.data_ptr()
        if sa1 is not None:
            a.a1 = sa1.data_ptr()
        touch = (sa0,) if sa1 is None else (sa0, sa1)
        scratch = torch.empty(a.M * max(a.ldc, 1) + 16 + (nb - 1) * a.out_bstride, device="cuda")
        a.out = scratch.data_ptr()
        if a.residual == saved[4]:
            a.residual = scratch.data_ptr()
        a.stats_out = 0
        if a.splitk > 1:
            a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
        base = time_call(lib, a, st, touch=touch)
        best = None
        tried = []
        if H2_MODE:
            a.a_split, a.a_split_ld = 0, 0
            ops.set_split_h2(a, H2_FLAG)
        else:
            ops.set_split(a)
        nkc = a.K // 32
        for cfg in ((1, 2, 4, 5) if H2_MODE else (1, 2, 4, 5, 21, 22)):          # 21 / 22: the warp-specialised 256x160 / 256x128 tiles (csrc/igemm_ws.hip)
            if a.epi == 1 and cfg not in EVEN_TN and cfg != 22:
                continue
            iters = -(-nkc // CFG_WK.get(cfg, 1))
            for sk in SKS:
                if sk > 1 and (a.epi == 1 or iters // sk < 1 or nb * sk * a.M * a.N > ws.numel()):
                    continue
                a.tile_cfg, a.splitk = cfg, sk
                a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
                t = time_call(lib, a, st, touch=touch)
                if t is not None:
                    tried.append((t, cfg, sk))
                if t is not None and (best is None or t < best[0]):
                    best = (t, cfg, sk)
        (a.tile_cfg, a.splitk, a.splitk_ws, a.splitk_ws_elems, a.out, a.stats_out, a.residual, a.compute, a.w_split, a.w_split_ld,
         a.w_split_bstride) = saved
        a.w_scale_exp, a.range_flag, a.a_split, a.a_split_ld = saved_h
        a.a0, a.a1 = saved_a
        if base is None or best is None:
            continue
        win = best[0] < 0.97 * base
        if win:      # runners-up of the winning arithmetic, for the in-step comparison (tools/instep_x3.sh)
            CANDS[key] = [[c, k, round(t, 5)] for t, c, k in sorted(set(tried))[:6]]
        if win:
            xtable[key] = [best[1], best[2]]
        elif key in xtable:
            del xtable[key]
        tag = "h2" if H2_MODE else "x3"
        print(f"{key:40s} f32 cfg={saved[0]} sk={saved[1]} {1e3 * base:8.1f} us | {tag} cfg={best[1]} sk={best[2]} {1e3 * best[0]:8.1f} us"
              f"  x{base / best[0]:.2f} {'-> ' + tag if win else ''}", flush=True)


def tune_program(pg, table):
    if X3_TABLE is not None:
        return tune_x3(pg, X3_TABLE)
    from dsml_thesis_amd.engine import plan_key
    lib = pg.lib
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(16 * 1024 * 1024 * 16, device="cuda")      # 1 GiB of split-K scratch for the search
    seen = {}
    for fn, args, a, name in pg.calls:
        if name != "ldmk_igemm":
            continue
        key = plan_key(a, a.M)
        rows_retune = ROWS_RETUNE and a.w_frag and a.a_mode == 0 and not a.b_trans
        slab_retune = SLAB_RETUNE and a.w_frag and not a.b_trans and max(1, a.batch) == 1
        if FORCE and key in table and key not in seen:
            del table[key]                      # --force: forget the recorded plan, sweep everything again
        if key in seen or (a.batch > 1 and not getattr(a, "_winograd", False)) or (key in table and not (rows_retune or slab_retune)):
            continue
        saved = (a.tile_cfg, a.splitk, a.splitk_ws, a.splitk_ws_elems, a.out, a.stats_out, a.residual)
        saved_a = (a.a0, a.a1)
        # stand-in A operands (finite random values) that time_call can rewrite between launches
        nb = max(1, a.batch)
        samples = -(-a.M // a.rows_per_sample)
        rows = samples * a.in_h * a.in_w if a.a_mode == 1 else a.M
        span = (nb - 1) * a.a_bstride if nb > 1 else 0
        sa0 = torch.randn(rows * a.c0 + span, device="cuda")
        sa1 = torch.randn(rows * a.c1 + span, device="cuda") if a.c1 else None
        a.a0 = sa0.data_ptr()
        if sa1 is not None:
            a.a1 = sa1.data_ptr()
        touch = (sa0,) if sa1 is None else (sa0, sa1)
        # time into scratch so that in-place residual / stats outputs of the real program are not disturbed
        scratch = torch.empty(a.M * max(a.ldc, 1) + 16 + (max(1, a.batch) - 1) * a.out_bstride, device="cuda")
        a.out = scratch.data_ptr()
        if a.residual == saved[4]:
            a.residual = scratch.data_ptr()
        a.stats_out = 0
        nkc = a.K // 32
        best = None
        tried = []
        if key in table:          # rows re-tune: the LDS-tiled plan on record is the one to beat
            a.tile_cfg, a.splitk = table[key]
            a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
            t = time_call(lib, a, st, touch=touch)
            if t is not None:             # (None: the plan on record cannot run this problem, e.g. a raw-slab GEMM of the small route)
                best = (t, a.tile_cfg, a.splitk)
                tried.append(best)
        if rows_retune:
            for cfg in range(7, 13):          # wave-autonomous row GEMM tiles (illegal ones return an error code)
                a.tile_cfg, a.splitk = cfg, 1
                t = time_call(lib, a, st, touch=touch)
                if t is not None:
                    tried.append((t, cfg, 1))
                if t is not None and (best is None or t < best[0]):
                    best = (t, cfg, 1)
        if slab_retune:
            small = bool(getattr(pg, "small_route", False))
            for cfg in range(13, 21):         # slab GEMM tiles (small row counts; illegal combinations return an error code)
                for sk in SKS:
                    if (a.epi == 1 and sk > 1 and not small) or sk * a.M * a.N > ws.numel():
                        continue
                    a.tile_cfg, a.splitk = cfg, sk
                    a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
                    geglu_split = a.epi == 1 and sk > 1   # small route: raw slabs + a GEGLU post launch (charged 4 us)
                    saved_gs = (a.raw_slabs, a.bias)
                    if geglu_split:
                        a.raw_slabs, a.bias = 1, 0
                    elif sk == 1 and a.raw_slabs:
                        a.raw_slabs = 0
                    t = time_call(lib, a, st, touch=touch)
                    a.raw_slabs, a.bias = saved_gs
                    if t is not None and geglu_split:
                        t += 0.004
                    if t is not None:
                        tried.append((t, cfg, sk))
                    if t is not None and (best is None or t < best[0]):
                        best = (t, cfg, sk)
        for cfg in range(1, 7):
            if key in table and best is not None:
                break
            if a.epi == 1 and cfg not in EVEN_TN:
                continue
            iters = -(-nkc // CFG_WK[cfg])
            for sk in SKS:
                if sk > 1 and (a.epi == 1 or iters // sk < 1 or max(1, a.batch) * sk * a.M * a.N > ws.numel()):
                    continue
                a.tile_cfg, a.splitk = cfg, sk
                a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
                t = time_call(lib, a, st, touch=touch)
                if t is not None:
                    tried.append((t, cfg, sk))
                if t is not None and (best is None or t < best[0]):
                    best = (t, cfg, sk)
        (a.tile_cfg, a.splitk, a.splitk_ws, a.splitk_ws_elems, a.out, a.stats_out, a.residual) = saved
        base = time_call(lib, a, st, touch=touch)
        a.a0, a.a1 = saved_a
        seen[key] = best
        table[key] = [best[1], best[2]]
        # runners-up (isolated timing, ms) for the in-step comparison: tools/instep_tune.py
        CANDS[key] = [[c, k, round(t, 5)] for t, c, k in sorted(set(tried))[:8]]
        print(f"{key:40s} heuristic cfg={saved[0]} sk={saved[1]} {1e3 * base:8.1f} us -> tuned cfg={best[1]} sk={best[2]} "
              f"{1e3 * best[0]:8.1f} us", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", action="append", default=[], help="[unet|dec|enc:]latent:batch, e.g. 64:16 or dec:64:16")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plans_f32.json"), help="flat table; tools/merge_plans.py --merge f32 puts it into the plan file")
    ap.add_argument("--fresh", action="store_true")
    ap.add_argument("--rows", action="store_true", help="re-tune rows-mode shapes already in the table against the "
                    "row-GEMM wave tiles (tile_cfg 7..12)")
    ap.add_argument("--slab", action="store_true", help="re-tune shapes already in the table against the slab-GEMM tiles "
                    "(tile_cfg 13..16, csrc/sgemm.hip): the small-batch cases")
    ap.add_argument("--force", action="store_true", help="re-sweep shapes that are already in the table (after a kernel change)")
    ap.add_argument("--x3", action="store_true", help="sweep the bf16x3 arithmetic (LDMK_COMPUTE_BF16X3) against the plans on record; "
                    "writes the flat table --x3-out (tools/merge_plans.py --merge bf16x3 / f16x2)")
    ap.add_argument("--x3-out", default=os.path.join(ROOT, "gpurun_out", "plans_bf16x3.json"))
    ap.add_argument("--h2", action="store_true", help="the --x3 sweep in the F16X2 arithmetic (LDMK_COMPUTE_F16X2, tile_cfg 1 / 2 / 4 / 5); "
                    "writes gpurun_out/plans_f16x2.json")
    a = ap.parse_args()
    if a.h2:
        a.x3, H2_MODE = True, True
        if a.x3_out.endswith("plans_bf16x3.json"):
            a.x3_out = a.x3_out.replace("plans_bf16x3.json", "plans_f16x2.json")
        os.environ["LDMK_H2_TABLE"] = "/nonexistent"
        H2_FLAG = torch.zeros(1, device="cuda", dtype=torch.int32)
    if a.x3:
        # the programs must be built with their f32 plans (table on, x3 table off) but with the split images packed
        os.environ["LDMK_X3_TABLE"] = "/nonexistent"
        from merge_plans import section as _section
        X3_TABLE = {} if a.fresh else (json.load(open(a.x3_out)) if os.path.exists(a.x3_out) else _section("f16x2" if a.h2 else "bf16x3"))
        for c in a.case or ["64:16", "32:16"]:
            parts = c.split(":")
            kind = parts[0] if len(parts) == 3 else "unet"
            lat, b = parts[-2:]
            print(f"== x3 {kind} latent {lat} batch {b}", flush=True)
            tune(kind, int(lat), int(b), None)
            json.dump(X3_TABLE, open(a.x3_out, "w"), indent=0, sort_keys=True)
            torch.cuda.empty_cache()
        json.dump(CANDS, open(a.x3_out + ".cands.json", "w"), indent=0, sort_keys=True)
        print(f"wrote {a.x3_out} ({len(X3_TABLE)} shapes)")
        sys.exit(0)
    ROWS_RETUNE = a.rows or a.force
    SLAB_RETUNE = a.slab
    FORCE = a.force
    table = {}
    if not a.fresh:
        from merge_plans import section as _section
        table = json.load(open(a.out)) if os.path.exists(a.out) else _section("f32")
    os.environ["LDMK_NO_PLAN_TABLE"] = "1"       # the search itself must start from the heuristic plans
    for c in a.case or ["64:16", "32:16"]:
        parts = c.split(":")
        kind = parts[0] if len(parts) == 3 else "unet"
        lat, b = parts[-2:]
        print(f"== {kind} latent {lat} batch {b}", flush=True)
        tune(kind, int(lat), int(b), table)
        json.dump(table, open(a.out, "w"), indent=0, sort_keys=True)
        torch.cuda.empty_cache()
    json.dump(table, open(a.out, "w"), indent=0, sort_keys=True)
    json.dump(CANDS, open(a.out + ".cands.json", "w"), indent=0, sort_keys=True)
    print(f"wrote {a.out} ({len(table)} shapes)")
