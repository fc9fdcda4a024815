"""Row GEMM vs LDS-tiled igemm on the UNet's transformer-block shapes (HIP events, median of N).

    python tools/rgemm_bench.py [--latent 64] [--batch 16]
Prints per shape: the best LDS-tiled (cfg, split-K) time from the plan table / heuristic and every legal row-GEMM tile."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsml_thesis_amd import lib as L  # noqa: E402
from dsml_thesis_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    shapes = []
    for lvl, c in enumerate((160, 320, 640)):
        hw = (a.latent >> lvl) ** 2
        m = a.batch * hw
        shapes += [(m, c, c, hw, "none", "bias"), (m, c, c, hw, "none", "bias+vec+res"), (m, c, c, hw, "none", "bias+res+stats"),
                   (m, c, c, hw, "affine", "bias"), (m, c, 3 * c, hw, "ln", "none"), (m, c, 8 * c, hw, "ln", "geglu"),
                   (m, 4 * c, c, hw, "none", "bias+res")]
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    for M, K, N, rps, pro, epi in shapes:
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        kw = {}
        if pro == "ln":
            kw.update(row_stats=ops.ln_stats(x), ln_gamma=torch.ones(K, device=dev), ln_beta=torch.zeros(K, device=dev))
        elif pro == "affine":
            kw.update(coef=torch.ones(M // rps, 2, K, device=dev))
        ncol = N
        if epi == "geglu":
            wp, bp = ops.pack_geglu(w, b)
            kw.update(geglu=True, bias=bp)
            ncol = N // 2
        else:
            wp = ops.pack_linear(w)
            kw.update(bias=b)
            if "vec" in epi:
                kw.update(batch_vec=torch.randn(M // rps, N, device=dev))
            if "res" in epi:
                kw.update(residual=torch.randn(M, N, device=dev))
            if "stats" in epi:
                kw.update(stats_out=torch.zeros(M // 32, N, 3, device=dev))
        wf = ops.pack_wfrag(wp)
        out = torch.empty(M, ncol, device=dev)
        gflop = 2.0 * M * N * K * 1e-9
        row = f"M={M:6d} K={K:5d} N={N:5d} {pro:6s} {epi:15s}"
        base = timeit(lambda: ops.linear(x, wp, rows_per_sample=rps, out=out, **kw))
        row += f" | igemm(auto) {base:7.1f} us {gflop / base * 1e3:6.1f} TF |"
        for cfg in range(7, 13):
            try:
                t = timeit(lambda: ops.linear(x, wp, rows_per_sample=rps, out=out, w_frag=wf, tile_cfg=cfg, **kw))
                row += f" c{cfg}:{t:6.1f}us/{gflop / t * 1e3:5.1f}TF"
            except L.LdmkError:
                row += f" c{cfg}:   --        "
        print(row, flush=True)


if __name__ == "__main__":
    main()
