"""BASELINE config 5 rehearsal on one GPU: UNet training step (p_losses forward + backward + AdamW + EMA), fp32.

  python tools/train_bench.py --batch 16 --latent 32 [--graph] [--steps 10]
Reports samples/s and the step's algorithmic TFLOP/s (3x the forward's GEMM FLOPs: forward + data-gradient + weight-
gradient products; the attention backward recomputes the scores, counted as 2.5x the forward attention FLOPs)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--latent", type=int, default=32, choices=[32, 64])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    from bench import build_model
    from dsml_thesis_amd.train import UNetTrainer
    dev = torch.device("cuda", 0)
    model, ucfg = build_model(a.latent, dev)
    unet = model.model.diffusion_model
    tr = UNetTrainer(unet)
    sa, sb = model.sqrt_alphas_cumprod, model.sqrt_one_minus_alphas_cumprod
    n, c, hw = a.batch, ucfg["in_channels"], a.latent
    g = torch.Generator(device="cpu").manual_seed(0)
    x0 = torch.randn(n, c, hw, hw, generator=g).to(dev)
    noise = torch.randn(n, ucfg["out_channels"], hw, hw, generator=g).to(dev)
    ctx = torch.randn(n, 1, ucfg["context_dim"], generator=g).to(dev)
    t = torch.randint(0, 1000, (n,), generator=g).to(dev)
    shadow = tr.P.flat.clone()
    loss_buf = torch.zeros(1, device=dev)

    def step():
        loss = tr.p_losses(x0, ctx, t, noise, sa, sb)
        tr.adamw_step(lr=1e-6)
        tr.ema_update(shadow, 0.9999)
        loss_buf.copy_(loss)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    run = step
    if a.graph:
        tr.P.step = 2          # the bias corrections are host scalars: freeze them for the captured replay
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        run = gr.replay
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fwd_gemm = (42.17 if a.latent == 32 else 168.62) * 1e9 * n
    fwd_attn = (3.84 if a.latent == 32 else 61.43) * 1e9 * n
    flops = 3 * fwd_gemm + 3.5 * fwd_attn
    print(json.dumps(dict(workload=f"UNet p_losses fwd+bwd+AdamW+EMA fp32, batch {n}, latent {hw}", graph=a.graph,
                          ms_per_step=round(dt * 1e3, 2), samples_per_s=round(n / dt, 2),
                          step_tflops=round(flops / dt / 1e12, 1), loss=float(loss_buf.item()),
                          peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))))


if __name__ == "__main__":
    main()
