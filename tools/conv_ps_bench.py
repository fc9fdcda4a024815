"""Route A/B for the ResBlock convolutions `GroupNorm -> SiLU -> Conv2d 3x3` of a UNet step, shape by shape, each as the launch
sequence NetBuilder.gn_conv emits (statistics finalize + apply / transform passes + GEMM + epilogue passes), four independent copies
per hipGraph so that no launch finds its operands in the L2 that just wrote them:
    default   what the step runs today: Winograd F(2x2,3x3) on pre-split planes (>= 320 channels, >= 256 tiles) or the LDS-tiled
              F16X2 implicit GEMM behind the fp32 gn_apply pass
    psc c,k   the direct convolution on the conv-mode pre-split tile (csrc/igemm_ps.hip: igemm_psc_kernel), tile_cfg c, split-K k
  python tools/conv_ps_bench.py [--latent 64] [--batch 16]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shapes_of(unet, H):
    """unique (c0, c1, cout, h) of the ResBlock convolutions, in execution order, with their multiplicity"""
    seen = {}
    h = H
    chans = []
    # replay the walk of UNetModel._build for the spatial sizes
    for i, blk in enumerate(unet.input_blocks):
        for m in blk.layers:
            if m.kind == "res":
                for key in ((m.cin, 0, m.cout, h), (m.cout, 0, m.cout, h)):
                    seen[key] = seen.get(key, 0) + 1
            elif m.kind == "down":
                h //= 2
        chans.append((blk.layers[-1].cout if blk.layers[-1].kind == "res" else getattr(blk.layers[-1], "ch", None), h))
    for m in unet.middle_block.layers:
        if m.kind == "res":
            for key in ((m.cin, 0, m.cout, h), (m.cout, 0, m.cout, h)):
                seen[key] = seen.get(key, 0) + 1
    for blk in unet.output_blocks:
        for m in blk.layers:
            if m.kind == "res":
                # the first convolution reads the skip concat: c0 = current stream, c1 = skip channels (sum = m.cin)
                seen[(m.cin, -1, m.cout, h)] = seen.get((m.cin, -1, m.cout, h), 0) + 1
                seen[(m.cout, 0, m.cout, h)] = seen.get((m.cout, 0, m.cout, h), 0) + 1
            elif m.kind == "up":
                h *= 2
    return seen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--copies", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    from dsml_thesis_amd import synth as W, ops
    from dsml_thesis_amd.engine import GraphedProgram, NetBuilder, Program
    from dsml_thesis_amd.unet import UNetModel
    cfg = W.NS_UNET if a.latent == 64 else W.FR_UNET
    shapes = shapes_of(UNetModel(**cfg), a.latent)
    n = a.batch
    dev = torch.device("cuda", 0)
    flag = torch.zeros(1, device=dev, dtype=torch.int32)
    g = torch.Generator(device="cuda").manual_seed(1)
    total = {}
    for (cin, c1, cout, h), mult in shapes.items():
        c0 = cin if c1 == 0 else (cin // 2) // 32 * 32          # (a two-source read of the skip concat; the split point does not matter here)
        c1 = cin - c0
        wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / np.sqrt(9 * cin)
        wp = ops.pack_conv3x3(wt)
        ops.pack_wsplit_h2(wp)
        wp_ps = ops.pack_wps(wp, h2=True)
        u = u_ps = None
        if cin >= NetBuilder.WINO_MIN_CIN:
            u = ops.pack_winograd(wt)
            ops.pack_wsplit_h2(u, batch=16)
            u_ps = ops.pack_wps(u, batch=16, h2=True)
        gamma, beta, bias = torch.ones(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(cout, device=dev)
        bv = torch.randn(n, cout, device=dev, generator=g)
        ks = [k for k in (1, 2, 4, 5) if (cin // 32) % k == 0] if n * h * h <= 16384 else [1]
        routes = [("default", None)] + [(f"psc {c},{k}", f"{c},{k}") for c in (23, 26, 27) for k in ks]
        res = {}
        for name, force in routes:
            if force is None:
                os.environ["LDMK_PSC"] = "0"
                os.environ.pop("LDMK_PSC_FORCE", None)
            else:
                os.environ["LDMK_PSC"] = "1"
                os.environ["LDMK_PSC_FORCE"] = force
            pg = Program(dev)
            pg.h2_flag = flag
            nb = NetBuilder(pg, n, None)
            try:
                for _ in range(a.copies):
                    x0 = pg.alloc(n, h, h, c0)
                    x0.copy_(torch.randn(n, h, h, c0, device=dev, generator=g))
                    x1 = None
                    if c1:
                        x1 = pg.alloc(n, h, h, c1)
                        x1.copy_(torch.randn(n, h, h, c1, device=dev, generator=g))
                    nb.gn_conv(x0, x1, h, h, gamma, beta, 1e-5, wp, u, bias, batch_vec=bv, bv_ld=cout, stats=True, u_ps=u_ps, wp_ps=wp_ps)
                gp = GraphedProgram(pg.run)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    gp.replay()
                e1.record()
                torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) * 1e3 / (a.reps * a.copies)
                if name == "default":
                    kinds = sorted(set(c[3] for c in pg.calls))
                    route = "winograd" if any("winograd" in k for k in kinds) else "direct"
            except Exception as e:       # an illegal (tile, split) for this shape
                res[name] = None
            del pg, nb
            torch.cuda.empty_cache()
        os.environ.pop("LDMK_PSC_FORCE", None)
        os.environ.pop("LDMK_PSC", None)
        flops = 2.0 * n * h * h * cout * 9 * cin
        best = min((v, k) for k, v in res.items() if v and k != "default")
        print(f"conv {c0:4d}+{c1:<4d}->{cout:4d} @{h:2d}x{h:<2d} x{mult:2d}  M={n * h * h:6d} K={9 * cin:5d} | default ({route}) {res['default']:7.1f} us "
              f"({flops / res['default'] / 1e6:5.1f} TF) | " + "  ".join(f"{k} {v:6.1f}" if v else f"{k} n/a" for k, v in res.items() if k != "default")
              + f" | best {best[1]} {best[0]:6.1f} us ({flops / best[0] / 1e6:5.1f} TF)  {res['default'] / best[0]:5.2f}x", flush=True)
        total["default"] = total.get("default", 0.0) + mult * res["default"]
        total["best"] = total.get("best", 0.0) + mult * min(best[0], res["default"])
        total["psc"] = total.get("psc", 0.0) + mult * best[0]
    print(f"per step (x multiplicity): default routes {total['default'] / 1e3:.3f} ms; all on the conv-mode tile {total['psc'] / 1e3:.3f} ms; "
          f"best of both per shape {total['best'] / 1e3:.3f} ms")
    assert int(flag.item()) == 0


if __name__ == "__main__":
    main()
