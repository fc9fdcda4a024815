"""Winograd F(2x2,3x3) against the direct implicit-GEMM convolution on the UNet's ResBlock shapes (HIP events, median).

    python tools/winograd_bench.py [--latent 64] [--batch 16]"""
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
    ap.add_argument("--upsample", action="store_true")
    a = ap.parse_args()
    if a.upsample:
        return upsample_main(a)
    n = a.batch
    for lvl, cin, cout in ((2, 640, 640), (2, 1280, 640), (2, 960, 640), (1, 320, 320), (1, 640, 320), (0, 160, 160)):
        h = a.latent >> lvl
        x = torch.randn(n, h, h, cin, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (9 * cin) ** 0.5
        b = torch.randn(cout, device="cuda")
        res = torch.randn(n, h, h, cout, device="cuda")
        coef = torch.ones(n, 2, cin, device="cuda")
        part = torch.zeros(n * h * h // 32, cout, 3, device="cuda")
        wp, u = ops.pack_conv3x3(w), ops.pack_winograd(w)
        xa = torch.empty_like(x)
        out = torch.empty(n, h, h, cout, device="cuda")
        tiles = n * (h // 2) * (h // 2)
        scratch = (torch.empty(16, tiles, cin, device="cuda"), torch.empty(16, tiles, cout, device="cuda"))

        def direct():
            L.call("ldmk_gn_apply", x.data_ptr(), cin, 0, 0, coef.data_ptr(), xa.data_ptr(), n, h * h, 1, ops.stream())
            ops.conv3x3(xa, wp, b, residual=res, out=out)

        def wino():
            ops.conv3x3_winograd(x, u, b, coef=coef, residual=res, out=out, stats_out=part, scratch=scratch)

        td, tw = timeit(direct), timeit(wino)
        gf = 2.0 * n * h * h * cout * 9 * cin * 1e-9
        print(f"{cin:5d}->{cout:4d} @{h:2d}x{h:<2d} n={n}: gn_apply + direct {td:7.1f} us ({gf / td * 1e3:6.1f} TF)   winograd {tw:7.1f} us "
              f"({gf / tw * 1e3:6.1f} TF direct-equivalent)  x{td / tw:.2f}", flush=True)


def upsample_main(a):
    n = a.batch
    for lvl, c in ((2, 640), (1, 320)):
        h = a.latent >> lvl
        x = torch.randn(n, h, h, c, device="cuda")
        w = torch.randn(c, c, 3, 3, device="cuda") / (9 * c) ** 0.5
        b = torch.randn(c, device="cuda")
        wp, w4 = ops.pack_conv3x3(w), ops.pack_upconv(w)
        out = torch.empty(n, 2 * h, 2 * h, c, device="cuda")
        part = torch.zeros(n * 4 * h * h // 32, c, 3, device="cuda")
        pix = n * h * h
        scratch = (torch.empty(4, pix, 4 * c, device="cuda"), torch.empty(4, pix, c, device="cuda"))
        td = timeit(lambda: ops.conv3x3(x, wp, b, upsample=True, out=out))
        tp = timeit(lambda: ops.upsample_conv3x3_phases(x, w4, b, out=out, stats_out=part, scratch=scratch))
        gf = 2.0 * n * 4 * h * h * c * 9 * c * 1e-9
        print(f"upsample {c}->{c} {h}x{h}->{2 * h}x{2 * h} n={n}: folded-gather conv {td:7.1f} us ({gf / td * 1e3:6.1f} TF)   four 2x2-tap phases "
              f"{tp:7.1f} us ({gf / tp * 1e3:6.1f} TF direct-equivalent)  x{td / tp:.2f}", flush=True)


if __name__ == "__main__":
    main()
