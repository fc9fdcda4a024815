"""Boundary convolutions (ldmk_conv3x3_in / ldmk_conv3x3_out) at the UNet and VQGAN shapes: python tools/boundary_conv_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsml_thesis_amd import ops
from tools.rgemm_bench import timeit
for (n, cin, cout, h) in [(16, 160, 4, 64), (16, 160, 3, 32), (1, 128, 3, 256), (16, 128, 3, 128)]:
    x = torch.randn(n, h, h, cin, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda"); b = torch.randn(cout, device="cuda")
    coef = torch.ones(n, 2, cin, device="cuda"); wp = ops.pack_conv3x3_narrow(w); out = torch.empty(n, cout, h, h, device="cuda")
    print(f"conv3x3_out n={n} {h}x{h}x{cin}->{cout}: {timeit(lambda: ops.conv3x3_out(x, coef, wp, b, cout, out=out)):7.1f} us")
for (n, cin, cout, h) in [(16, 4, 160, 64), (16, 3, 160, 32), (1, 4, 512, 64), (16, 3, 128, 128)]:
    x = torch.randn(n, cin, h, h, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda"); b = torch.randn(cout, device="cuda")
    wp = ops.pack_conv3x3_narrow(w); out = torch.empty(n, h, h, cout, device="cuda")
    print(f"conv3x3_in  n={n} {h}x{h}x{cin}->{cout}: {timeit(lambda: ops.conv3x3_in(x, wp, b, cout, out=out)):7.1f} us")
