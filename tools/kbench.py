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
    if what == "geglu":
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
