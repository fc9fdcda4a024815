"""Diagnostic: where a slab-GEMM launch (csrc/sgemm.hip) spends its time.  Builds a second copy of the library with
-DLDMK_SG_STAMPS (per-wave s_memrealtime stamps, 100 MHz), runs one batch-1 problem per configuration and prints the phase
timeline across all waves of the launch.
    python tools/sgemm_probe.py --build          (container: cross-compiles tools/bin/libldmk_sgprobe.so)
    python tools/sgemm_probe.py                  (GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "bin", "libldmk_sgprobe.so")


def build():
    sys.path.insert(0, ROOT)
    from dsml_thesis_amd import build as B
    B.build_lib(verbose=False)
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    obj = os.path.join(ROOT, "tools", "bin", "sgemm_probe.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DLDMK_SG_STAMPS", "-c", os.path.join(B.CSRC, "sgemm.hip"), "-o", obj])
    objs = [os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != "sgemm.hip"] + [obj]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    print("built", SO)


def main():
    os.environ["LDMK_LIBRARY"] = SO
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from dsml_thesis_amd import lib as L
    from dsml_thesis_amd import ops
    L.load()
    L.init(0)
    NW = {13: 4, 14: 4, 15: 4, 16: 4, 17: 8, 18: 16, 19: 8, 20: 8}
    TILE = {13: (2, 1), 14: (2, 2), 15: (1, 1), 16: (1, 2), 17: (1, 1), 18: (1, 1), 19: (2, 1), 20: (1, 2)}
    cases = [("conv 64x640x5760", 1, 640, 640, 8, 8, True), ("rows 64x640x640", 1, 640, 640, 8, 8, False),
             ("conv 1024x160x1440", 1, 160, 160, 32, 32, True), ("rows 1024x160x160", 1, 160, 160, 32, 32, False)]
    for name, n, cin, cout, h, w, conv in cases:
        M = n * h * w
        x = torch.randn(n, h, w, cin, device="cuda")
        if conv:
            wp = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)
            K = 9 * cin
        else:
            wp = ops.pack_linear(torch.randn(cout, cin, device="cuda") * 0.05)
            K = cin
        wf = ops.pack_wfrag(wp)
        out = torch.empty(M, cout, device="cuda")
        ws = torch.empty(64 * M * cout, device="cuda")
        for cfg, sk in ((15, 6), (18, 6), (18, 3), (17, 6), (13, 6), (15, 12)):
            if NW[cfg] * sk > K // 8:
                continue
            tm, tn = TILE[cfg]
            tiles = (M // (32 * tm)) * (cout // (32 * tn))
            nwav = tiles * sk * NW[cfg]
            stamps = torch.zeros(nwav * 8, dtype=torch.int64, device="cuda")
            a = ops.make_igemm_args(M, cout, K, x, cin, wp, out, cout, h * w, conv=(h, w, h, w, 1, 1, 0) if conv else None,
                                    w_frag=wf, tile_cfg=cfg, splitk=sk, splitk_ws=ws, raw_slabs=True)
            a.stats_out = stamps.data_ptr()
            junk = torch.empty(64 * 1024 * 1024, device="cuda")
            for rep in range(3):
                junk.fill_(rep)                      # evict the weights from L2 / Infinity Cache like the rest of a step does
                stamps.zero_()
                torch.cuda.synchronize()
                ops.igemm(a)
                torch.cuda.synchronize()
            t = stamps.view(nwav, 8).cpu().numpy().astype(np.float64) * 0.01      # us
            t0 = t[:, 0].min()
            w0 = t[np.arange(nwav) % NW[cfg] == 0]                                 # waves that store
            ph = lambda i, j, tt=t: np.median(tt[:, j] - tt[:, i])
            print(f"{name:20s} cfg={cfg} sk={sk:2d} {nwav:5d} waves | first..last wave start {t[:, 0].max() - t0:5.2f} us | "
                  f"setup {ph(0, 1):4.2f}  issue {ph(1, 2):4.2f}  loop {ph(2, 3):5.2f} (max {np.max(t[:, 3] - t[:, 2]):5.2f})  "
                  f"tree {ph(3, 4):4.2f}  store {ph(4, 5, w0):4.2f} | span {w0[:, 5].max() - t0:6.2f} us", flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
