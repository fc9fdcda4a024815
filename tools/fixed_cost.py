"""Where a sampling call's time goes besides its DDIM steps (one GPU, FR config at 64x64x4 or 32x32x3, B = 16):
pack_weights, ema_scope enter / exit, the first sample() (programs + graph capture), later sample() calls at DDIM-50 / DDIM-200
against the pure replay rate of the captured step over the same number of steps.   python tools/fixed_cost.py [--latent 64]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tic():
    torch.cuda.synchronize()
    return time.perf_counter()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    from bench import StepRunner, build_model
    from dsml_thesis_amd.ddim import DDIMSampler
    dev = torch.device("cuda", 0)
    model, ucfg = build_model(a.latent, dev)
    unet = model.model.diffusion_model
    t0 = tic()
    unet.pack_weights()
    print(f"pack_weights (all weight forms): {tic() - t0:.3f} s")
    labels = (torch.arange(a.batch, device=dev) % 8)[:, None]
    shape = [ucfg["in_channels"], a.latent, a.latent]
    t0 = tic()
    cm = model.ema_scope()
    cm.__enter__()
    t1 = tic()
    c = model.cond_stage_model.embedding(labels)
    s = DDIMSampler(model)
    for S in (50, 200, 200, 50):
        t2 = tic()
        s.sample(S=S, batch_size=a.batch, shape=shape, conditioning=c, eta=0.0, verbose=False, use_graph=True)
        el = tic() - t2
        print(f"sample(S={S}) inside ema_scope: {el:.3f} s = {a.batch * S / el:.1f} sample-steps/s")
    t2 = tic()
    s2 = DDIMSampler(model)
    s2.sample(S=200, batch_size=a.batch, shape=shape, conditioning=c, eta=0.0, verbose=False, use_graph=True)
    print(f"sample(S=200) from a NEW DDIMSampler object: {tic() - t2:.3f} s")
    t2 = tic()
    cm.__exit__(None, None, None)
    t3 = tic()
    print(f"ema_scope enter {t1 - t0:.3f} s, exit {t3 - t2:.3f} s")
    t2 = tic()
    with model.ema_scope():
        s.sample(S=50, batch_size=a.batch, shape=shape, conditioning=c, eta=0.0, verbose=False, use_graph=True)
    el = tic() - t2
    print(f"`with ema_scope(): sample(S=50)` the second time, enter + sample + exit: {el:.3f} s = {a.batch * 50 / el:.1f} sample-steps/s")
    run = StepRunner(model, ucfg, a.batch, graph=True)
    for n in (20, 50, 200):
        for _ in range(3):
            run.step()
        t2 = tic()
        for _ in range(n):
            run.step()
        el = tic() - t2
        print(f"pure replay of the captured step, {n} steps: {1e3 * el / n:.3f} ms/step = {a.batch * n / el:.1f} sample-steps/s")
