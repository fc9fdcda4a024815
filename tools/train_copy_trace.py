"""Diagnostic: which Python lines of the training step issue device-to-device copies (they show up as
__amd_rocclr_copyBuffer in the rocprof summary).  One eager p_losses under torch.profiler with stacks.
    python tools/train_copy_trace.py --latent 32 --batch 4
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--bf16", action="store_true")
    a = ap.parse_args()
    from bench import build_model
    from dsml_thesis_amd.train import UNetTrainer
    dev = torch.device("cuda", 0)
    model, ucfg = build_model(a.latent, dev)
    tr = UNetTrainer(model.model.diffusion_model, compute="bf16" if a.bf16 else "f32")
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(a.batch, ucfg["in_channels"], a.latent, a.latent, generator=g).to(dev)
    noise = torch.randn(a.batch, ucfg["out_channels"], a.latent, a.latent, generator=g).to(dev)
    ctx = torch.randn(a.batch, 1, ucfg["context_dim"], generator=g).to(dev)
    t = torch.randint(0, 1000, (a.batch,), generator=g).to(dev)
    sa, sb = model.sqrt_alphas_cumprod, model.sqrt_one_minus_alphas_cumprod
    tr.p_losses(x0, ctx, t, noise, sa, sb)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.p_losses(x0, ctx, t, noise, sa, sb)
        torch.cuda.synchronize()
    by_site = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zero_", "aten::fill_", "aten::cat"):
            site = next((s for s in ev.stack if "dsml_thesis_amd" in s), "?")
            by_site[(ev.name, site)] += 1
    for (name, site), n in by_site.most_common(30):
        print(f"{n:5d}  {name:18s} {site}")
    kinds = collections.Counter(ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    for name, n in kinds.most_common(12):
        print(f"{n:5d}  device: {name[:100]}")


if __name__ == "__main__":
    main()
