"""Diagnostic: what a small dependent libldmk launch costs inside a hipGraph, and why it is 4.8 us in the batch-1 step
when a chain of identical copy kernels costs 1.8 us (tools/launch_floor.hip).  Chains of N launches, timed by replaying
the captured graph:  same kernel + same buffers / rotating buffers / alternating kernels / small kernels between GEMMs.
    python tools/floor_lib.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsml_thesis_amd import lib as L  # noqa: E402
from dsml_thesis_amd import ops  # noqa: E402
from dsml_thesis_amd.engine import GraphedProgram  # noqa: E402


def timed(fn, n_launch, reps=20):
    g = GraphedProgram(fn)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n_launch


def main():
    dev = torch.device("cuda", 0)
    lib = L.load()
    L.init(0)
    st = lambda: torch.cuda.current_stream().cuda_stream
    N = 200
    rows, C = 64, 640
    pool = torch.randn(N, rows * C + 4096, device=dev)          # rotating 170 KB buffers
    x = pool[0, :rows * C]
    stats = torch.empty(N, rows, 2, device=dev)
    temb = torch.empty(1, 160, device=dev)
    t_in = torch.zeros(1, dtype=torch.int64, device=dev)
    freqs = ops.timestep_freqs(160, device=dev)
    hw = 64
    chunks = lib.ldmk_gn_chunks(hw)
    part = torch.randn(chunks, C, 3, device=dev)
    coef = torch.empty(N, 2, C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y = torch.empty(N, rows * C, device=dev)

    def ln_same():
        for i in range(N):
            L.check(lib.ldmk_ln_stats(x.data_ptr(), rows, C, 1e-5, stats[0].data_ptr(), st()))

    def ln_rot():
        for i in range(N):
            L.check(lib.ldmk_ln_stats(pool[i].data_ptr(), rows, C, 1e-5, stats[i].data_ptr(), st()))

    def temb_same():
        for i in range(N):
            L.check(lib.ldmk_timestep_embedding(t_in.data_ptr(), freqs.data_ptr(), temb.data_ptr(), 1, 160, st()))

    def fin_same():
        for i in range(N):
            L.check(lib.ldmk_gn_finalize(part.data_ptr(), C, 0, 0, 1, hw, 32, 1e-5, gamma.data_ptr(), beta.data_ptr(),
                                         coef[0].data_ptr(), st()))

    def apply_same():
        for i in range(N):
            L.check(lib.ldmk_gn_apply(x.data_ptr(), C, 0, 0, coef[0].data_ptr(), y[0].data_ptr(), 1, hw, 1, st()))

    def mix4():
        for i in range(N // 4):
            L.check(lib.ldmk_gn_finalize(part.data_ptr(), C, 0, 0, 1, hw, 32, 1e-5, gamma.data_ptr(), beta.data_ptr(),
                                         coef[i].data_ptr(), st()))
            L.check(lib.ldmk_gn_apply(pool[i].data_ptr(), C, 0, 0, coef[i].data_ptr(), y[i].data_ptr(), 1, hw, 1, st()))
            L.check(lib.ldmk_ln_stats(y[i].data_ptr(), rows, C, 1e-5, stats[i].data_ptr(), st()))
            L.check(lib.ldmk_timestep_embedding(t_in.data_ptr(), freqs.data_ptr(), temb.data_ptr(), 1, 160, st()))

    # a GEMM between the small kernels: M=64 N=640 K=640, LDS-tiled 64x128 tile, split-K 6 (two kernels per call)
    w = torch.randn(640, 640, device=dev) * 0.05
    wp = ops.pack_linear(w)
    outs = torch.empty(N, rows, C, device=dev)
    ws = torch.empty(6 * rows * C, device=dev)
    import ctypes as Cc

    def gemm_args(i):
        a = ops.make_igemm_args(rows, C, C, pool[i, :rows * C].view(rows, C), C, wp, outs[i], C, hw)
        a.tile_cfg, a.splitk = 4, 6
        a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
        return a
    gargs = [gemm_args(i) for i in range(N)]

    def gemm_only():
        for i in range(N):
            L.check(lib.ldmk_igemm(Cc.byref(gargs[i]), st()))

    def gemm_ln():
        for i in range(N):
            L.check(lib.ldmk_igemm(Cc.byref(gargs[i]), st()))
            L.check(lib.ldmk_ln_stats(outs[i].data_ptr(), rows, C, 1e-5, stats[i].data_ptr(), st()))

    print(f"ln_stats x{N}, same buffers:            {timed(ln_same, N):6.2f} us per launch")
    print(f"ln_stats x{N}, rotating buffers:        {timed(ln_rot, N):6.2f} us per launch")
    print(f"timestep_embedding x{N}:                {timed(temb_same, N):6.2f} us per launch")
    print(f"gn_finalize x{N}, same buffers:         {timed(fin_same, N):6.2f} us per launch")
    print(f"gn_apply x{N}, same buffers:            {timed(apply_same, N):6.2f} us per launch")
    print(f"finalize/apply/ln_stats/temb mix:       {timed(mix4, N):6.2f} us per launch")
    tg = timed(gemm_only, N)
    tgl = timed(gemm_ln, N)
    print(f"igemm 64x640x640 sk=6 (+reduce) x{N}:    {tg:6.2f} us per call (2 kernels)")
    print(f"igemm + ln_stats x{N}:                  {tgl:6.2f} us per pair -> ln_stats adds {tgl - tg:5.2f} us")


if __name__ == "__main__":
    main()
