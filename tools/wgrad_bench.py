"""Time ldmk_wgrad on the UNet's weight-gradient shapes (HIP events, median of 10): python tools/wgrad_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsml_thesis_amd import train_ops as T    # noqa: E402

CONV = [(16, 32, 160, 160), (16, 32, 320, 160), (16, 16, 320, 320), (16, 16, 640, 320), (16, 8, 640, 640), (16, 8, 1280, 640),
        (16, 32, 480, 160)]
LIN = [(16384, 160, 480), (16384, 160, 1280), (16384, 640, 160), (16384, 160, 160), (4096, 320, 960), (4096, 320, 2560),
       (4096, 1280, 320), (1024, 640, 1920), (1024, 640, 5120), (1024, 2560, 640)]


def med(fn, reps=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    tot = 0.0
    for n, h, cin, cout in CONV:
        x = torch.randn(n, h, h, cin, device="cuda")
        dy = torch.randn(n, h, h, cout, device="cuda")
        dw = torch.empty(9 * cin, cout, device="cuda")
        db = torch.empty(cout, device="cuda")
        t = med(lambda: T.wgrad_conv3x3(x, dy, dw=dw, dbias=db))
        fl = 2.0 * n * h * h * 9 * cin * cout
        tot += t
        print(f"conv  n={n} hw={h} cin={cin} cout={cout}: {t * 1e3:8.1f} us  {fl / t / 1e9:6.1f} TFLOP/s")
    for R, K, N in LIN:
        a = torch.randn(R, K, device="cuda")
        dy = torch.randn(R, N, device="cuda")
        dw = torch.empty(K, N, device="cuda")
        db = torch.empty(N, device="cuda")
        t = med(lambda: T.wgrad_linear(a, dy, dw=dw, dbias=db))
        tot += t
        print(f"lin   R={R} K={K} N={N}: {t * 1e3:8.1f} us  {2.0 * R * K * N / t / 1e9:6.1f} TFLOP/s")
    print(f"total {tot * 1e3:.1f} us")


if __name__ == "__main__":
    main()
