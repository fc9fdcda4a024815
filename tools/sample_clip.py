"""BASELINE configs 3/4: talking-face clip generation (audio-conditioned LDM, DDIM-200) on synthetic inputs.

  python tools/sample_clip.py --frames 128 --mode fixed            # whole clip as one batch, hipGraph step
  python tools/sample_clip.py --frames 16 --mode autoreg           # the reference's serial chain (B=1)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/sample_clip.py --frames 128
      -> 16 frames per GPU, one RCCL all-gather of the decoded frames at the end (config 4)

Prints one JSON line with frames/s and sample-steps/s (whole job, decode and gather included).
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
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--mode", choices=["fixed", "autoreg"], default="fixed")
    ap.add_argument("--window", type=int, default=8)
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LDMK_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    from dsml_thesis_amd.synth import make_tf_model
    from dsml_thesis_amd.ddim import DDIMSampler, C12, C34
    from dsml_thesis_amd import ops
    from dsml_thesis_amd.parallel import sample_sharded
    T, W_ = a.frames, a.window
    m = make_tf_model(gain=0.25, seq_len=2 * W_ + 1, device=dev)
    rs = np.random.RandomState(2)
    audio = torch.from_numpy(rs.standard_normal((T, 768)).astype(np.float32)).to(dev)
    masked = torch.tanh(torch.from_numpy(rs.standard_normal((T, 3, 128, 128)).astype(np.float32))).to(dev)
    masked[:, :, 70:, :] = -1.0
    ident = torch.tanh(torch.from_numpy(rs.standard_normal((1, 3, 128, 128)).astype(np.float32))).to(dev)
    c1 = m.cond_stage_model_1.embedding(torch.tensor([[4]], device=dev))
    s = DDIMSampler(m)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if a.mode == "autoreg":
        assert world == 1, "the autoregressive chain is serial: replicas only (different clips per GPU)"
        xid = m.encode_first_stage(ident)
        frames, _ = s.progressive_sampling(c1, xid, masked, audio, a.steps, 1, T, [3, 32, 32], W_, eta=0.0, verbose=False)
        out = ops.postprocess_frames(m.decode_first_stage(torch.cat(frames)))
    else:
        xid = m.encode_first_stage(ident)
        idx = torch.tensor([[min(max(f + i, 0), T - 1) for i in range(-W_, W_ + 1)] for f in range(T)], device=dev)

        def cond(lo, hi):
            c2 = m.cond_stage_model_2(audio[idx[lo:hi]])
            c12 = torch.cat([c1.expand(hi - lo, -1, -1), c2], dim=2)
            c34 = torch.cat([m.encode_first_stage(masked[lo:hi]), xid.expand(hi - lo, -1, -1, -1)], dim=1)
            return {C12: c12, C34: c34}
        out = sample_sharded(s, a.steps, T, (3, 32, 32), cond, seed=5, rank=rank, world_size=world)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    assert out.shape == (T, 128, 128, 3) and torch.isfinite(out).all()
    if rank == 0:
        print(json.dumps(dict(workload=f"talking_face clip, {T} frames, DDIM-{a.steps}, mode={a.mode}", n_gpus=world,
                              seconds=round(el, 3), frames_per_s=round(T / el, 3),
                              sample_steps_per_s=round(T * a.steps / el, 1), checksum=float(out.double().sum()))))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
