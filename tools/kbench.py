"""Micro-benchmarks of single kernels on the bench shapes (HIP-event timing, median of interleaved rounds)."""
import sys
import os
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsml_thesis_amd import ops, lib as L  # noqa: E402


def timeit(fn, reps=int(os.environ.get('KB_REPS', '20')), warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def attn(n, tokens, heads):
    C = heads * 32
    qkv = torch.randn(n * tokens, 3 * C, device="cuda")
    out = torch.empty(n * tokens, C, device="cuda")
    us = timeit(lambda: ops.attn_self(qkv, n, tokens, heads, out=out))
    fl = 4.0 * n * heads * tokens * tokens * 32
    print(f"attn_self n={n} tokens={tokens} heads={heads}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")


def gemm(M, N, K, conv=None, cfg=0, sk=0, ln=False, geglu=False, res=False):
    cin = K // 9 if conv else K
    if conv:
        n, h, w = conv
        x = torch.randn(n, h, w, cin, device="cuda")
    else:
        x = torch.randn(M, K, device="cuda")
    wt = torch.randn(K, N, device="cuda") / K ** 0.5
    out = torch.empty(M, N // 2 if geglu else N, device="cuda")
    ws = torch.empty(8 * M * N, device="cuda")
    kw = {}
    if ln:
        kw = dict(tf=L.TF_LAYERNORM, row_stats=torch.randn(M, 2, device="cuda").abs() + 0.5,
                  ln_gamma=torch.ones(K, device="cuda"), ln_beta=torch.zeros(K, device="cuda"))
    r = torch.randn(M, N, device="cuda") if res else None
    a = ops.make_igemm_args(M, N, K, x, cin, wt, out, out.shape[1], (conv[1] * conv[2]) if conv else M,
                            conv=(conv[1], conv[2], conv[1], conv[2], 1, 1, 0) if conv else None,
                            epi=L.EPI_GEGLU if geglu else L.EPI_NONE, residual=r, splitk_ws=ws, splitk=sk, **kw)
    a.tile_cfg = cfg
    us = timeit(lambda: ops.igemm(a))
    fl = 2.0 * M * N * K
    print(f"igemm M={M} N={N} K={K} conv={bool(conv)} cfg={cfg} sk={sk} ln={ln} geglu={geglu} res={res}: "
          f"{us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "attn"
    if what == "hbm":
        n, hw, c = 16, 4096, 320
        x = torch.randn(n * hw, c, device="cuda")
        g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        coef = ops.gn_coef(x, None, n, hw, g, b, 1e-5)
        y = torch.empty_like(x)
        us = timeit(lambda: ops.gn_apply(x, None, coef, n, hw, out=y))
        print(f"gn_apply {n}x{hw}x{c}: {us:8.1f} us  {2 * x.numel() * 4 / us / 1e3:7.1f} GB/s (read+write)")
        part = torch.empty(n * (hw // 32) * c * 3, device="cuda")
        us = timeit(lambda: L.call("ldmk_gn_partial", x.data_ptr(), c, n, hw, part.data_ptr(), ops.stream()))
        print(f"gn_partial {n}x{hw}x{c}: {us:8.1f} us  {x.numel() * 4 / us / 1e3:7.1f} GB/s (read)")
        x2 = torch.randn(n * hw, 160, device="cuda")
        st = torch.empty(n * hw, 2, device="cuda")
        us = timeit(lambda: ops.ln_stats(x2, out=st))
        print(f"ln_stats {n * hw}x160: {us:8.1f} us  {x2.numel() * 4 / us / 1e3:7.1f} GB/s (read)")
        z = torch.randn(128, 3, 128, 128, device="cuda")
        o = torch.empty(128, 128, 128, 3, device="cuda")
        us = timeit(lambda: ops.postprocess_frames(z, out=o))
        print(f"postprocess_frames 128x3x128x128: {us:8.1f} us  {2 * z.numel() * 4 / us / 1e3:7.1f} GB/s (read+write)")
        cb = torch.randn(16384, 3, device="cuda")
        zz = torch.randn(128, 3, 32, 32, device="cuda")
        us = timeit(lambda: ops.vq_nearest(zz, cb))
        print(f"vq_nearest 128x1024 vectors vs 16384 codes: {us:8.1f} us  {128 * 1024 * 16384 / us / 1e6:7.2f} T distance-evals/s")
    elif what == "geglu":
        for M, N, K in ((65536, 1280, 160), (4096, 5120, 640), (16384, 2560, 320)):
            for ln in (False, True):
                for gg in (False, True):
                    gemm(M, N, K, ln=ln, geglu=gg)
            for cfg in (1, 2, 5):
                gemm(M, N, K, cfg=cfg)
    elif what == "attnqt":
        for qt in (1, 2):
            L.load().ldmk_attn_force_qt(qt)
            print("qt", qt)
            attn(16, 4096, 5)
            attn(16, 1024, 10)
            attn(16, 1024, 5)
            attn(16, 256, 20)
    elif what == "attn":
        attn(16, 4096, 5)
        attn(16, 1024, 10)
        attn(16, 256, 20)
        attn(16, 1024, 5)
        attn(16, 64, 20)
    elif what == "pmc":
        gemm(65536, 160, 1440, conv=(16, 64, 64))
        gemm(65536, 160, 2880, conv=(16, 64, 64))
        gemm(4096, 5120, 640, ln=True, geglu=True)
        gemm(65536, 160, 160, res=True)
        attn(16, 4096, 5)
    elif what == "gemm":
        gemm(65536, 160, 1440, conv=(16, 64, 64))
        gemm(4096, 5120, 640, ln=True, geglu=True)
        gemm(65536, 1280, 160, ln=True, geglu=True)
        gemm(65536, 160, 160, res=True)
        gemm(65536, 160, 640, res=True)
        gemm(16384, 160, 160, res=True)
        gemm(1024, 640, 5760, conv=(16, 8, 8))
        for cfg in (1, 2, 4, 5, 6):
            gemm(16384, 160, 160, cfg=cfg, res=True)
        for cfg in (1, 2, 5):
            gemm(4096, 1920, 640, cfg=cfg, ln=True)
