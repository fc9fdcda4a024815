"""s_memtime stamps of the warp-specialised pre-split GEMM (csrc/igemm_ps.hip, LDMK_PS_DEBUG=8): where a stage's cycles go, per role.
   LDMK_HIPCC_FLAGS=-DLDMK_PS_PROBES python -m dsml_thesis_amd.build && LDMK_PS_DEBUG=8 LDMK_HIPCC_FLAGS=-DLDMK_PS_PROBES python tools/pw_stamps.py
   (the probes exist only in builds made with -DLDMK_PS_PROBES)"""
import os
import sys
import numpy as np
import torch

os.environ.setdefault("LDMK_PS_DEBUG", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsml_thesis_amd import lib as L, ops  # noqa: E402

for (M, N, K, B, cfg, label) in [(1024, 640, 640, 16, 29, "Winograd 640 @16 x16"), (65536, 160, 1440, 1, 29, "conv as rows"),
                                 (4096, 1920, 640, 1, 29, "QKV L2"), (65536, 480, 160, 1, 29, "QKV L0"), (4096, 5120, 640, 1, 30, "GEGLU L2 tile")]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, M, K, device="cuda", generator=g)
    w = (torch.randn(B, K, N, device="cuda", generator=g) / np.sqrt(K)).contiguous()
    wp = w if B > 1 else w[0].contiguous()
    wps, xps = ops.pack_wps(wp, batch=B), ops.pack_ps(x if B > 1 else x[0])
    out = torch.empty(B, M, N, device="cuda")
    tiles = ((M + 255) // 256) * ((N + (159 if cfg == 29 else 127)) // (160 if cfg == 29 else 128))
    dbgbuf = torch.zeros(tiles * 8 * 4, dtype=torch.int64, device="cuda")
    kw = dict(batch=B, w_bstride=K * N, out_bstride=M * N) if B > 1 else {}
    a = ops.make_igemm_args(M, N, K, None, K, wp, out, N, M, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=wps, **kw)
    a.splitk_counters, a.splitk_counters_len = dbgbuf.data_ptr(), dbgbuf.numel()
    if "--col" in sys.argv and B == 1:         # the lane = column epilogue (chosen when GroupNorm records are asked for)
        rec = torch.zeros(M // 32, N, 3, device="cuda")
        a.stats_out = rec.data_ptr()
    for _ in range(3):
        ops.igemm(a)
    torch.cuda.synchronize()
    d = dbgbuf.cpu().numpy().reshape(tiles, 8, 4).astype(np.float64)
    n16 = K // 16
    cons, prod = d[:, :4].mean((0, 1)), d[:, 4:].mean((0, 1))
    print(f"{label}: M={M} N={N} K={K} x{B}, {n16} stages; per STAGE cycles (total / stages)")
    print(f"  consumer: compute {cons[0] / n16:7.0f}  barrier wait {cons[2] / n16:7.0f}  epilogue(total) {cons[1]:8.0f}  total {cons[3]:9.0f}  (matrix work alone: {n16 * 1920})")
    print(f"  producer: issue   {prod[0] / n16:7.0f}  vmcnt wait   {prod[1] / n16:7.0f}  barrier wait {prod[2] / n16:7.0f}  total {prod[3]:9.0f}")
