"""One GEMM shape / tile configuration, launched repeatedly (for rocprofv3 --pmc / --kernel-trace runs).
    python tools/rgemm_one.py M K N cfg [iters] [epi]      cfg 0 = LDS-tiled heuristic, 7..12 = row-GEMM wave tiles"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsml_thesis_amd import ops  # noqa: E402

M, K, N, cfg = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
epi = sys.argv[6] if len(sys.argv) > 6 else "bias"
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") / K ** 0.5
wp = ops.pack_linear(w)
wf = ops.pack_wfrag(wp)
out = torch.empty(M, N, device="cuda")
kw = dict(bias=torch.randn(N, device="cuda"))
if "res" in epi:
    kw["residual"] = torch.randn(M, N, device="cuda")
for _ in range(iters):
    ops.linear(x, wp, rows_per_sample=M, out=out, w_frag=wf if cfg >= 7 else None, tile_cfg=cfg, **kw)
torch.cuda.synchronize()
print("done")
