"""BASELINE config 2 end to end (the native counterpart of face_reenactment/sample_affectnet.py:66-137):
class-conditional faces, DDIM-S, optional CFG, decode, clamp to [0,1], NHWC, save .npy.

  python tools/sample_faces.py --n 16 --steps 200 --scale 3.0 --out faces.npy
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--latent", type=int, default=32, choices=[32, 64])
    ap.add_argument("--out", default=None)
    ap.add_argument("--repeat", type=int, default=2, help="runs; the last one (programs built, graphs captured) is reported")
    a = ap.parse_args()
    from bench import build_model
    from dsml_thesis_amd.ddim import DDIMSampler
    from dsml_thesis_amd import ops
    model, ucfg = build_model(a.latent, torch.device("cuda", 0))
    sampler = DDIMSampler(model)
    labels = (torch.arange(a.n, device="cuda") % 8)[:, None]
    with model.ema_scope():                                                    # sample_affectnet.py:86
        c = model.cond_stage_model.embedding(labels)                           # :108-109
        uc = model.cond_stage_model.uncond_embedding(torch.zeros_like(labels)) if a.scale > 1.0 else None   # :93-94
        for _ in range(max(1, a.repeat)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            z, _ = sampler.sample(S=a.steps, batch_size=a.n, shape=[ucfg["in_channels"], a.latent, a.latent],
                                  conditioning=c, eta=a.eta, unconditional_guidance_scale=a.scale,
                                  unconditional_conditioning=uc, verbose=False, use_graph=True)      # :117
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            x = ops.postprocess_frames(model.decode_first_stage(z))            # :126-132
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        el, ts, td = t2 - t0, t1 - t0, t2 - t1
    if a.out:
        np.save(a.out, x.cpu().numpy())
    print(json.dumps(dict(workload=f"class-conditional faces n={a.n} DDIM-{a.steps} scale={a.scale} latent={a.latent}",
                          seconds=round(el, 3), frames_per_s=round(a.n / el, 3), sampling_seconds=round(ts, 3),
                          sample_steps_per_s=round(a.n * a.steps / ts, 1), unet_evals_per_sample_step=2 if a.scale > 1.0 else 1,
                          decode_seconds=round(td, 4), decode_frames_per_s=round(a.n / td, 1), shape=list(x.shape),
                          finite=bool(torch.isfinite(x).all()))))


if __name__ == "__main__":
    main()
