"""Diagnostic for the attention tail (64x64x4, batch 16: 2560 workgroups of 128 queries on 1024 workgroup slots = 2.5 rounds):
the same product with 128-query (QT = 1) and 256-query (QT = 2: 1280 workgroups, 2 per CU resident = 2.5 rounds of 512) tiles,
and on batches that make the workgroup count a whole number of rounds -- per-call time and TFLOP/s, HIP events, median of 20.
    python tools/attn_rounds.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dsml_thesis_amd import lib as L  # noqa: E402
from dsml_thesis_amd import ops  # noqa: E402
from rgemm_bench import timeit  # noqa: E402


def main():
    lib = L.load()
    tokens, heads = 4096, 5
    for batch in (16, 13, 19, 26, 32):          # 13 x 5 x 32 = 2080 ~ 2.03 rounds; 19 -> 3040 (2.97); 26 -> 4160 (4.06)
        qkv = torch.randn(batch * tokens, 3 * heads * 32, device="cuda")
        out = torch.empty(batch * tokens, heads * 32, device="cuda")
        gf = 4.0 * tokens * tokens * 32 * heads * batch * 1e-9
        for qt in (1, 2):
            lib.ldmk_attn_force_qt(qt)
            t = timeit(lambda: ops.attn_self(qkv, batch, tokens, heads, out=out))
            wgs = batch * heads * (tokens // (128 * qt))
            print(f"batch {batch:2d} QT={qt}: {wgs:5d} workgroups = {wgs / 1024:5.2f} rounds of 1024 slots | {t:8.1f} us "
                  f"{gf / t * 1e3:6.1f} TFLOP/s = {gf / t * 1e3 / 157.3:5.3f} of peak", flush=True)
        lib.ldmk_attn_force_qt(0)


if __name__ == "__main__":
    main()
