"""3x3 convolution on the LDS-tiled igemm (128x160 tile) at 1 / 2 / 4 workgroups per CU: python tools/conv_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsml_thesis_amd import ops, lib
from tools.rgemm_bench import timeit
for cin, cout, hw in [(160, 160, 64), (320, 160, 64), (320, 320, 32), (640, 640, 16)]:
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (9 * cin) ** 0.5
    wp = ops.pack_conv3x3(w)
    for n in (4, 8, 16, 32):
        x = torch.randn(n, hw, hw, cin, device="cuda")
        out = torch.empty(n, hw, hw, cout, device="cuda")
        M = n * hw * hw
        tiles = (M // 128) * (cout // 160 if cout % 160 == 0 else -(-cout // 128))
        lib.load().ldmk_igemm_force_config(5)
        t = timeit(lambda: ops.conv3x3(x, wp, None, out=out))
        lib.load().ldmk_igemm_force_config(0)
        fl = 2.0 * M * cout * 9 * cin
        print(f"conv {cin}->{cout} @{hw}x{hw} n={n:2d} M={M:6d} tiles(128x160)={tiles:4d}: {t:8.1f} us {fl / t * 1e-6:6.1f} TF")
