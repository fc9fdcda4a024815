"""Which arithmetic every GEMM launch of a UNet program runs in, and which sites (if any) lost F16X2 to a range flag.
   python tools/plan_report.py --latent 32 --batch 7 [--policy 16]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--policy", type=int, default=None)
    a = ap.parse_args()
    from dsml_thesis_amd import synth as W, lib as L
    unet_cfg = W.NS_UNET if a.latent == 64 else W.FR_UNET
    m = W.make_fr_model(gain=0.25, unet=unet_cfg, vq=W.VQ_F4_256 if a.latent == 64 else W.VQ_F4, device="cuda")
    unet = m.model.diffusion_model
    unet.policy_batch = a.policy
    g = torch.Generator().manual_seed(0)
    x = torch.randn(a.batch, unet_cfg["in_channels"], a.latent, a.latent, generator=g).cuda()
    t = (torch.arange(a.batch) * 37 % 1000).cuda()
    c = m.cond_stage_model.embedding((torch.arange(a.batch, device="cuda") % 8)[:, None])
    import warnings
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        e = unet(x, t, context=c)
    for w in rec:
        print("warning:", str(w.message)[:300])
    st = unet.arithmetic_status()
    print(f"latent {a.latent} B = {a.batch} policy {a.policy}: finite {bool(torch.isfinite(e).all())}; sites {st['sites']}, denied {len(st['denied'])}: {st['denied'][:12]}")
    pg = unet.program(a.batch, a.latent, a.latent, 1, 0)
    names = {L.COMPUTE_F32: "f32", L.COMPUTE_BF16: "bf16", L.COMPUTE_BF16X3: "bf16x3", L.COMPUTE_F16X2: "f16x2"}
    cnt = collections.Counter()
    rows = collections.Counter()
    for _, _, ar, name in pg.calls:
        if name != "ldmk_igemm":
            continue
        cnt[names[ar.compute]] += 1
        if ar.compute in (L.COMPUTE_F32,):
            rows[f"M={ar.M} N={ar.N} K={ar.K} mode={ar.a_mode} tf={ar.a_tf} epi={ar.epi} batch={ar.batch} cfg={ar.tile_cfg} sk={ar.splitk}"] += 1
    print("GEMM launches by arithmetic:", dict(cnt))
    for k, v in sorted(rows.items()):
        print(f"  f32 x{v}: {k}")
    att = collections.Counter(n for _, _, _, n in pg.calls if "attn" in n)
    print("attention launches:", dict(att))
