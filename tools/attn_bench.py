"""Self-attention kernel alone on the UNet's three levels (HIP events, median of 20).

    python tools/attn_bench.py [--latent 64] [--batch 16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dsml_thesis_amd import ops  # noqa: E402
from rgemm_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lib", help="another build of libldmk.so to time instead (A/B on one box)")
    ap.add_argument("--mode", default="f32", choices=("f32", "x3", "x3p", "h2"), help="f32 matrix cores / bf16x3 / bf16x3 with the K, V pre-pass")
    a = ap.parse_args()
    if a.lib:
        from dsml_thesis_amd import lib as L
        L.LIB_PATH = os.path.abspath(a.lib)
    tot = 0.0
    for lvl, (heads, calls) in enumerate(((5, 5), (10, 5), (20, 6))):
        tokens = (a.latent >> lvl) ** 2
        torch.manual_seed(17 + lvl)
        qkv = torch.randn(a.batch * tokens, 3 * heads * 32, device="cuda")
        out = torch.empty(a.batch * tokens, heads * 32, device="cuda")
        if a.mode == "x3p":
            from dsml_thesis_amd import lib as L
            kv = torch.empty(L.load().ldmk_attn_kv_split_bytes(a.batch, tokens, heads), device="cuda", dtype=torch.uint8)
            t = timeit(lambda: L.call("ldmk_attn_self_x3p", qkv.data_ptr(), kv.data_ptr(), out.data_ptr(), a.batch, tokens, heads, 32 ** -0.5, ops.stream()))
        elif a.mode == "h2":
            from dsml_thesis_amd import lib as L
            kv = torch.empty(L.load().ldmk_attn_kv_split_h2_bytes(a.batch, tokens, heads), device="cuda", dtype=torch.uint8)
            flag = torch.zeros(1, device="cuda", dtype=torch.int32)
            t = timeit(lambda: L.call("ldmk_attn_self_h2", qkv.data_ptr(), kv.data_ptr(), out.data_ptr(), flag.data_ptr(), a.batch, tokens, heads,
                                      32 ** -0.5, ops.stream()))
            assert int(flag.item()) == 0
            nb = min(a.batch, 2)         # accuracy against float64 on the first two samples
            q, k, v = (t.reshape(nb, tokens, heads, 32).permute(0, 2, 1, 3).double() for t in qkv[:nb * tokens].split(heads * 32, dim=1))
            ref = (torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, dim=-1) @ v).permute(0, 2, 1, 3).reshape(nb * tokens, heads * 32)
            err = (out[:nb * tokens].double() - ref).abs().max().item()
            print(f"   max |error| against float64 {err:.3e} (max |ref| {ref.abs().max().item():.3f})")
            import hashlib          # (two builds on one box: the digests say whether the results are the same bits)
            print("   digest out", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16], "kv", hashlib.sha1(kv.cpu().numpy().tobytes()).hexdigest()[:16])
        else:
            t = timeit(lambda: ops.attn_self(qkv, a.batch, tokens, heads, out=out, x3=a.mode == "x3"))
        gf = 4.0 * tokens * tokens * 32 * heads * a.batch * 1e-9
        tot += t * calls
        print(f"tokens {tokens:5d} heads {heads:2d}: {t:8.1f} us  {gf / t * 1e3:6.1f} TFLOP/s  (x{calls} per step)", flush=True)
    print(f"per step: {tot / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
