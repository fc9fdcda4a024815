"""What the LayerNorm prologue costs inside the token-row GEMMs: same shape and epilogue, with and without it.

    python tools/ln_cost.py [--latent 64] [--batch 16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dsml_thesis_amd import lib as L  # noqa: E402
from dsml_thesis_amd import ops  # noqa: E402
from rgemm_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cfgs", action="store_true", help="print every tile, not only the best")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    for lvl, c in enumerate((160, 320, 640)):
        hw = (a.latent >> lvl) ** 2
        M = a.batch * hw
        for N, epi in ((3 * c, "none"), (8 * c, "geglu")):
            K = c
            x = torch.randn(M, K, generator=g).to(dev)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
            b = torch.randn(N, generator=g).to(dev)
            ln = dict(row_stats=ops.ln_stats(x), ln_gamma=torch.ones(K, device=dev), ln_beta=torch.zeros(K, device=dev))
            kw = {}
            ncol = N
            if epi == "geglu":
                wp, bp = ops.pack_geglu(w, b)
                kw.update(geglu=True, bias=bp)
                ncol = N // 2
            else:
                wp = ops.pack_linear(w)
            wf = ops.pack_wfrag(wp)
            out = torch.empty(M, ncol, device=dev)
            gflop = 2.0 * M * N * K * 1e-9
            row = f"M={M:6d} K={K:4d} N={N:5d} {epi:6s}"
            w2, cs, b2 = ops.fold_layernorm(wp, ln["ln_gamma"], ln["ln_beta"], kw.get("bias"))
            wf2 = ops.pack_wfrag(w2)
            folded = dict(row_stats=ln["row_stats"], ln_colsum=cs)
            for name, extra in (("ln", ln), ("folded", folded), ("raw", {})):
                best = None
                for cfg in list(range(1, 7)) + list(range(7, 13)):
                    try:
                        if name == "folded":
                            k2 = dict(kw, bias=b2)
                            t = timeit(lambda: ops.linear(x, w2, rows_per_sample=hw, out=out, w_frag=wf2 if cfg > 6 else None,
                                                          tile_cfg=cfg, **k2, **extra))
                        else:
                            t = timeit(lambda: ops.linear(x, wp, rows_per_sample=hw, out=out, w_frag=wf if cfg > 6 else None,
                                                          tile_cfg=cfg, **kw, **extra))
                    except L.LdmkError:
                        continue
                    if best is None or t < best[0]:
                        best = (t, cfg)
                row += f" | {name}: {best[0]:7.1f} us c{best[1]:<2d} {gflop / best[0] * 1e3:6.1f} TF"
            print(row, flush=True)


if __name__ == "__main__":
    main()
