"""Isolated timing of the pre-split bf16x3 tiles (csrc/igemm_ps.hip, tile_cfg 23..28) against the LDS-tiled / warp-specialised
bf16x3 forms on the GEMM shapes of the 64x64x4, B = 16 step.  Operand copies are rotated between launches so that no launch
finds its A rows in the L2 that just read them (as inside the step).   python tools/ps_bench.py [--quick]"""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsml_thesis_amd import lib as L, ops  # noqa: E402

SHAPES = [  # (M, N, K, geglu, lnf, batch, label)
    (65536, 160, 1440, False, False, 1, "conv 160->160 @64 (as rows)"),
    (65536, 1280, 160, True, True, 1, "GEGLU L0"),
    (16384, 2560, 320, True, True, 1, "GEGLU L1"),
    (4096, 5120, 640, True, True, 1, "GEGLU L2"),
    (65536, 480, 160, False, True, 1, "QKV L0"),
    (16384, 960, 320, False, True, 1, "QKV L1"),
    (4096, 1920, 640, False, True, 1, "QKV L2"),
    (65536, 160, 640, False, False, 1, "ff2 L0"),
    (16384, 320, 1280, False, False, 1, "ff2 L1"),
    (4096, 640, 2560, False, False, 1, "ff2 L2"),
    (4096, 640, 640, False, False, 1, "to_out L2"),
    (16384, 320, 320, False, False, 1, "to_out L1"),
    (65536, 160, 160, False, False, 1, "to_out L0"),
    (1024, 640, 640, False, False, 16, "Winograd 640 @16 x16"),
    (4096, 320, 320, False, False, 16, "Winograd 320 @32 x16"),
    (16384, 160, 320, False, False, 16, "Winograd 320->160 @64 x16"),
]


def timeit(fn, n_rot, reps=12):
    for i in range(3):
        fn(i % n_rot)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % n_rot)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    quick = "--quick" in sys.argv
    probe = "--probe" in sys.argv          # the shapes tools/ps_probe.sh looks at: long K, short K + heavy epilogue, L2-resident batch
    h2 = "--h2" in sys.argv                # the same comparison in the F16X2 arithmetic (two fp16 planes; tile_cfg 29 / 30 do not exist there)
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    COMP = L.COMPUTE_F16X2 if h2 else L.COMPUTE_BF16X3
    shapes = SHAPES[:6] if quick else ([SHAPES[0], SHAPES[1], SHAPES[4], SHAPES[13]] if probe else SHAPES)
    mdiv = 4 if "--latent32" in sys.argv else 1      # the same layers at 32x32 (a quarter of the rows)
    for (M, N, K, geglu, lnf, B, label) in shapes:
        M //= mdiv
        R = 3
        g = torch.Generator(device="cuda").manual_seed(1)
        xs = [torch.randn(B, M, K, device="cuda", generator=g) for _ in range(R)]
        w = (torch.randn(B, K, N, device="cuda", generator=g) / np.sqrt(K)).contiguous()
        wp = w if B > 1 else w[0].contiguous()
        (ops.pack_wsplit_h2 if h2 else ops.pack_wsplit)(wp, batch=B)
        wps = ops.pack_wps(wp, batch=B, h2=h2)
        xps = [ops.pack_ps(x if B > 1 else x[0], h2_flag=flag if h2 else None) for x in xs]
        ncol = N // 2 if geglu else N
        out = torch.empty(B, M, ncol, device="cuda")
        st = torch.stack([torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")], 1).contiguous() if lnf else None
        cs = torch.zeros(N, device="cuda") if lnf else None
        bias = torch.zeros(N, device="cuda")
        ws = torch.empty(8 * B * M * N, device="cuda") if M * N * B <= 64 * 1024 * 1024 else None
        kw = dict(bias=bias, epi=L.EPI_GEGLU if geglu else L.EPI_NONE, tf=L.TF_LAYERNORM_FOLDED if lnf else L.TF_NONE,
                  row_stats=st, ln_colsum=cs)
        if B > 1:
            kw.update(batch=B, w_bstride=K * N, out_bstride=M * ncol)
        flops = 2.0 * B * M * N * K
        res = {}
        cands = [(1 if geglu else 5, 1)] if probe else [(1 if geglu else 5, 1), (22 if geglu else 21, 1)]
        pcands = [(25, 1), (28, 1), (30, 1), (32, 1), (33, 1)] if geglu else [(23, 1), (24, 1), (26, 1), (27, 1), (29, 1), (31, 1)]
        if not geglu and M * N <= 4096 * 640 and K >= 640:
            cands += [(5, 2), (5, 4)]
            pcands += [(23, 2), (23, 4), (27, 2), (27, 4), (29, 2), (29, 4), (31, 2), (31, 4)]
        if h2:
            cands = [c for c in cands if c[0] <= 6] + ([(1, 1)] if not geglu else [])
            pcands = [c for c in pcands if c[0] not in (29, 30)]
        for cfg, sk in cands:
            args = []
            for r in range(R):
                a = ops.make_igemm_args(M, N, K, xs[r], K, wp, out, ncol, M, tile_cfg=cfg, splitk=sk, splitk_ws=ws if sk > 1 else None,
                                        compute=COMP, range_flag=flag, a_bstride=M * K if B > 1 else 0, **kw)
                args.append(a)
            try:
                res[f"x3 cfg{cfg} sk{sk}"] = timeit(lambda r: ops.igemm(args[r]), R)
            except L.LdmkError as e:
                res[f"x3 cfg{cfg} sk{sk}"] = None
        for cfg, sk in pcands:
            for ops_out in ([False, True] if geglu else [False]):
                args = []
                ops_ps = ops.ps_empty(M, ncol, h2=h2) if ops_out else None
                for r in range(R):
                    a = ops.make_igemm_args(M, N, K, None, K, wp, None if ops_out else out, ncol, M, tile_cfg=cfg, splitk=sk,
                                            splitk_ws=ws if sk > 1 else None, a_ps=xps[r], w_ps=wps, out_ps=ops_ps, range_flag=flag, **kw)
                    args.append(a)
                try:
                    res[f"ps cfg{cfg} sk{sk}" + (" ->ps" if ops_out else "")] = timeit(lambda r: ops.igemm(args[r]), R)
                except L.LdmkError as e:
                    res[f"ps cfg{cfg} sk{sk}"] = None
        best_x3 = min(v for k, v in res.items() if k.startswith("x3") and v)
        best_ps = min(v for k, v in res.items() if k.startswith("ps") and v)
        line = f"{label:28s} M={M:6d} N={N:5d} K={K:5d} x{B:<2d} | " + "  ".join(
            f"{k} {v:7.1f}us {flops / v / 1e6:6.1f}TF" if v else f"{k} n/a" for k, v in res.items())
        print(line, flush=True)
        print(f"{'':28s} best x3 {best_x3:7.1f} us ({flops / best_x3 / 1e6:6.1f} TF)   best ps {best_ps:7.1f} us ({flops / best_ps / 1e6:6.1f} TF)   "
              f"speed-up {best_x3 / best_ps:5.2f}x", flush=True)
        del xs, xps, out, ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
