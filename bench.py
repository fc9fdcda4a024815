#!/usr/bin/env python
"""bench.py -- DDIM denoise throughput of the MI355X-native latent-diffusion sampling path.

    python bench.py --gpus N --steps K --warmup W
        N > 1: run it as is -- the parent process (before it touches a GPU) starts the N ranks itself with
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` and relays
        their JSON line -- or launch it through torch.distributed.run yourself (the driver does): both work.

A "step" is one DDIM denoise step (conditioned UNet evaluation + fused DDIM update) over one batch of
latents -- BASELINE.json configs[1]: face_reenactment AffectNet emotion-conditioned LDM, DDIM-200 schedule,
batch 16 per GPU, at the latent size BASELINE.json's metric is quoted on (64x64x4; `--latent 32` selects the
shipped 32x32x3 shape, which is also measured briefly and reported under "secondary").  Weights are random
(seeded recipe, dsml_thesis_amd/synth.py: no checkpoints exist offline), inputs synthetic and resident in HBM before
the timed region.  Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- the GEMM kernel family (LDS-tiled igemm + row GEMM, f32 MFMA): the FLOPs its launches EXECUTE in one
                  step (sum of 2 M N K over the launch program) divided by their summed durations, measured with HIP
                  events on the launch stream; the reference algorithm's FLOPs are printed beside it
  clip         -- BASELINE configs[2]/[3]: the 128-frame talking-face clip (fixed identity) sharded over the ranks, ONE
                  all-gather of the decoded frames inside the timed region, checksum identical for every N
  cpu_baseline -- the oracle (PyTorch-CPU fp32 restatement of the reference) on this box's host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic work per sample-step (forward hooks on the reference, 2*MAC)
GFLOP_STEP = {32: 46.011, 64: 230.051}
GFLOP_IGEMM = {32: 24.318 + 15.122 + 2.726, 64: 97.297 + 60.421 + 10.905}   # conv3x3 + linear + conv1x1
# algorithmic bytes of one step: the fp32 weights read once (627.0 MB) + the in / out of every ResBlock and
# SpatialTransformer per sample (28 MB at 32x32x3, 111 MB at 64x64x4)
ALGO_BYTES = {32: lambda b: int(627.0e6 + b * 28e6), 64: lambda b: int(627.0e6 + b * 111e6)}
PEAK_F32_MFMA = 157.3  # TFLOP/s, MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32, dense)
PEAK_BF16_MFMA = 2516.6  # TFLOP/s dense = 16 x the f32 matrix rate (same guide: "1/16 of BF16 MFMA", ~2.5 PF)


def build_model(latent, device):
    from dsml_thesis_amd import synth as W
    unet = W.NS_UNET if latent == 64 else W.FR_UNET
    vq = W.VQ_F4_256 if latent == 64 else W.VQ_F4
    return W.make_fr_model(gain=0.25, unet=unet, vq=vq, device=device), unet


class StepRunner:
    """Owns the device-resident DDIM loop state for one batch and advances it one step per call."""

    def __init__(self, model, unet_cfg, batch, ddim_steps=200, graph=True, seed=1):
        from dsml_thesis_amd.ddim import DDIMSampler
        from dsml_thesis_amd import lib as L
        from dsml_thesis_amd.engine import GraphedProgram
        self.L = L
        dev = model.device
        self.model, self.batch = model, batch
        s = DDIMSampler(model)
        s.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        self.sampler = s
        C_, H = unet_cfg["in_channels"], unet_cfg["image_size"]
        labels = (torch.arange(batch, device=dev) % 8)[:, None]
        ctx = model.cond_stage_model.embedding(labels)                                # (B,1,512)
        unet = model.model.diffusion_model
        unet.auto_repack = True
        self.pg = pg = unet.program(batch, H, H, 1, 0)
        unet.auto_repack = False
        x_T = torch.from_numpy(np.random.RandomState(seed).standard_normal((batch, C_, H, H)).astype(np.float32)).to(dev)
        self.x_T = x_T
        pg.inputs["context"].copy_(ctx.reshape(batch, -1))
        pg.ctx_program.run()
        self.S = ddim_steps
        self.step_idx = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pred_x0 = torch.empty_like(x_T)
        self.per = x_T[0].numel()
        self.reset()
        self.graph = None
        if graph:
            self.graph = GraphedProgram(self.eager_step)
            self.reset()

    def reset(self):
        self.pg.inputs["x"].copy_(self.x_T)
        self.step_idx.fill_(self.S - 1)
        self.pg.inputs["t"].fill_(int(self.sampler.ddim_timesteps[self.S - 1]))

    def eager_step(self):
        pg, s = self.pg, self.sampler
        pg.run()
        x = pg.inputs["x"]
        rc = pg.lib.ldmk_ddim_step(x.data_ptr(), pg.outputs["eps"].data_ptr(), 0, s._table.data_ptr(),
                                   self.step_idx.data_ptr(), 1.0, 0, x.data_ptr(), self.pred_x0.data_ptr(), self.per,
                                   self.batch, s._ts_table.data_ptr(), pg.inputs["t"].data_ptr(), self.batch, 1, self.S,
                                   torch.cuda.current_stream().cuda_stream)
        self.L.check(rc, "ldmk_ddim_step")

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.eager_step()

    def igemm_time_per_step(self, reps=3):
        """Sum of igemm launch durations in one step, from HIP events recorded on the launch stream."""
        pg = self.pg
        st = torch.cuda.current_stream().cuda_stream
        ig = pg.lib.ldmk_igemm
        n_ig = sum(1 for c in pg.calls if c[3] == "ldmk_igemm")
        best = None
        for _ in range(reps):
            evs, tevs = [], []
            for fn, args, _, name in pg.calls:
                if name in ("ldmk_igemm", "ldmk_winograd_input", "ldmk_winograd_input_ps", "ldmk_winograd_input_ps_h2", "ldmk_winograd_output",
                            "ldmk_upconv_gather", "ldmk_upconv_gather_ps", "ldmk_upconv_gather_ps_h2", "ldmk_upconv_scatter"):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = fn(*args, st)
                    e1.record()
                    (evs if name == "ldmk_igemm" else tevs).append((e0, e1))
                else:
                    rc = fn(*args, st)
                self.L.check(rc, name)
            torch.cuda.synchronize()
            tot = sum(a.elapsed_time(b) for a, b in evs)
            if best is None or tot < best:
                best = tot
                self.winograd_transform_ms = sum(a.elapsed_time(b) for a, b in tevs)
                # the same launches by arithmetic: LDMK_COMPUTE_BF16X3 (six bf16 MFMAs per fp32-accurate product) vs f32 MFMA
                comp = [c[2].compute for c in pg.calls if c[3] == "ldmk_igemm"]
                self.x3_ms = sum(a.elapsed_time(b) for (a, b), f in zip(evs, comp) if f == 2)
                self.x3_launches = sum(1 for f in comp if f == 2)
                # ... and LDMK_COMPUTE_F16X2 (three fp16 MFMAs per product)
                self.h2_ms = sum(a.elapsed_time(b) for (a, b), f in zip(evs, comp) if f == 3)
                self.h2_launches = sum(1 for f in comp if f == 3)
        return best, n_ig


def sustained_mfma16_tflops():
    """Median TFLOP/s of the register-only v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands (profiles/r04_clock_bf16.txt)."""
    import re
    try:
        vals = []
        mode = None
        for ln in open(os.path.join(ROOT, "profiles", "r04_clock_bf16.txt")):
            m = re.search(r"MFMAs per wave: [\d.]+ ms = ([\d.]+) TFLOP/s; shader clock median (\d+) MHz", ln)
            if m and float(m.group(1)) > 400 and int(m.group(2)) < 2000:      # (bf16 rows on real operands: the zero-operand runs hold > 2.1 GHz)
                vals.append(float(m.group(1)))
        vals.sort()
        return round(vals[len(vals) // 2], 1) if vals else None
    except Exception:
        return None


def executed_gemm_flops(pg, compute=None):
    """fp32-equivalent FLOPs of the GEMM launches of one step: sum of 2 M N K (x batch) over the launch program (compute:
    only the launches of that arithmetic).  A bf16x3 launch issues SIX bf16 MFMA FLOPs per FLOP counted here."""
    fl = 0.0
    for _, _, a, name in pg.calls:
        if name == "ldmk_igemm" and (compute is None or a.compute == compute):
            fl += 2.0 * a.M * a.N * a.K * max(1, a.batch)
    return fl


def algorithmic_gemm_flops(pg):
    """FLOPs of the Conv2d / Linear layers those launches compute, in the reference's arithmetic: 2 M N K, and for a
    convolution that runs through Winograd F(2x2,3x3) (16 batched GEMMs over M/4 tiles) or as four 2x2-tap phase convolutions
    (nearest-x2 upsampling + conv) the direct form's 2 M N (9 C)."""
    fl = 0.0
    for _, _, a, name in pg.calls:
        if name == "ldmk_igemm":
            fl += getattr(a, "_algo_flops", 2.0 * a.M * a.N * a.K * max(1, a.batch))
    return fl


def host_cores():
    """CPU cores this process may actually use (cgroup quota / affinity), not the host's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16 if n > 64 else n))   # a 1-GPU box's CPU share is 16 cores (task statement)


def cpu_baseline(latent, budget_s=30.0, min_steps=20):
    """The oracle restatement on the host cores: bounded sample of the same workload (rank 0, N=1 only).  Every DDIM
    step is timed on its own and the MEDIAN step time is reported (three single-shot values on record differed by
    +-25 %)."""
    from oracle import ldm_oracle as O
    from oracle import weights as W
    cfg = W.NS_UNET if latent == 64 else W.FR_UNET
    torch.set_num_threads(host_cores())
    sd = W.synth_state_dict(W.unet_param_shapes(cfg), gain=0.25)
    b = 2
    x = torch.randn(b, cfg["in_channels"], latent, latent)
    ctx = torch.randn(b, 1, 512)
    sched = O.register_schedule(**W.SCHEDULE)
    ts = O.make_ddim_timesteps(200)
    tab = O.make_ddim_tables(sched["alphas_cumprod"], ts, 0.0)
    times, t_all = [], time.perf_counter()
    with torch.no_grad():
        for n in range(min_steps + 1):
            idx = 199 - n
            t0 = time.perf_counter()
            t = torch.full((b,), int(ts[idx]), dtype=torch.long)
            e = O.apply_model(sd, cfg, x, t, [ctx])
            x, _ = O.ddim_update(x, e, tab["a_t"][idx], tab["a_prev"][idx], tab["sigma_t"][idx],
                                 tab["sqrt_one_minus_at"][idx])
            if n > 0:                       # the first step pays one-off allocation / thread-pool start-up
                times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget_s and len(times) >= 5:
                break
    med = float(np.median(times))
    return dict(value=round(b / med, 3), unit="sample-steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"median of {len(times)} DDIM steps of the oracle (PyTorch-CPU fp32 restatement), B={b}, "
                       f"{latent}x{latent}x{cfg['in_channels']} latent, FR UNet; step times {min(times):.2f}-{max(times):.2f} s")


def clip_leg(rank, world, dev, dist, barrier, frames=128, ddim_steps=20, window=8, policy="shard"):
    """BASELINE configs[2] (N = 1) / configs[3] (N > 1): the talking-face clip in fixed-identity mode, frames
    sharded contiguously over the ranks, per-frame start noise from (seed, global frame index), hipGraph-captured DDIM
    step, decode, and ONE all-gather of the decoded frames -- all inside the timed region.  policy "shard" (the default of
    the timed leg): every rank runs the launch plans tuned for ITS frames/GPU batch, result equal to the 1-GPU run within the
    sampling tolerance; policy "job" (--clip-policy job): plans pinned to the whole clip, gathered result and checksum bitwise
    independent of N (what the rank-count test compares)."""
    from dsml_thesis_amd.synth import make_tf_model
    from dsml_thesis_amd.ddim import DDIMSampler, C12, C34
    from dsml_thesis_amd.parallel import sample_sharded
    T, W_ = frames, window
    m = make_tf_model(gain=0.25, seq_len=2 * W_ + 1, device=dev)
    rs = np.random.RandomState(2)
    audio = torch.from_numpy(rs.standard_normal((T, 768)).astype(np.float32)).to(dev)
    masked = torch.tanh(torch.from_numpy(rs.standard_normal((T, 3, 128, 128)).astype(np.float32))).to(dev)
    masked[:, :, 70:, :] = -1.0
    ident = torch.tanh(torch.from_numpy(rs.standard_normal((1, 3, 128, 128)).astype(np.float32))).to(dev)
    c1 = m.cond_stage_model_1.embedding(torch.tensor([[4]], device=dev))
    s = DDIMSampler(m)
    idx = torch.tensor([[min(max(f + i, 0), T - 1) for i in range(-W_, W_ + 1)] for f in range(T)], device=dev)

    def job():
        xid = m.encode_first_stage(ident)

        def cond(lo, hi):
            c2 = m.cond_stage_model_2(audio[idx[lo:hi]])
            c12 = torch.cat([c1.expand(hi - lo, -1, -1), c2], dim=2)
            c34 = torch.cat([m.encode_first_stage(masked[lo:hi]), xid.expand(hi - lo, -1, -1, -1)], dim=1)
            return {C12: c12, C34: c34}
        return sample_sharded(s, ddim_steps, T, (3, 32, 32), cond, seed=5, rank=rank, world_size=world, policy=policy)

    from dsml_thesis_amd import parallel as _par
    job()                                   # builds the launch programs and captures the step graph (untimed)
    issued0 = _par.COLLECTIVES_ISSUED
    els = []
    njobs = 1 if ddim_steps >= 100 else 3   # the shipped DDIM-200 job takes seconds: one timed job; short ones: the median of three
    for _ in range(njobs):                  # whole jobs, each between barriers (a single SHORT job varied by +-7 % from run to
        barrier()                           # run: host-side start noise, allocator state)
        t0 = time.perf_counter()
        out = job()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        els.append(el)
    el = sorted(els)[len(els) // 2]
    assert out.shape == (T, 128, 128, 3) and torch.isfinite(out).all()
    per = -(-T // world)
    return {"workload": f"talking_face audio-conditioned LDM, {T}-frame clip (fixed identity), DDIM-{ddim_steps}, encode + "
                        f"{ddim_steps} steps + decode + gather, {per} frames/GPU", "frames": T, "ddim_steps": ddim_steps,
            "scaling": "strong", "plan_policy": policy, "seconds": round(el, 4), "seconds_of_jobs": [round(e, 4) for e in els],
            "frames_per_s": round(T / el, 2),
            "sample_steps_per_s": round(T * ddim_steps / el, 1),
            "collective": ("none (1 rank)" if _par.COLLECTIVES_ISSUED == issued0 else
                           f"one all_gather_into_tensor of the decoded frames per job, {per}x128x128x3 fp32 per rank"),
            "collectives_issued_in_timed_jobs": _par.COLLECTIVES_ISSUED - issued0,
            "ranks_seen": (dist.get_world_size() if dist is not None else 1),
            "backend": (dist.get_backend() if dist is not None else None),
            "checksum": float(out.double().sum())}


def shard_cost_leg(dev, ddim_steps, frames=16, job_frames=128, window=8):
    """One rank's share of configs[3] on this one GPU: 16 frames of the 128-frame clip, DDIM steps + decode, timed with the
    launch plans of the whole job (policy "job": bitwise equal to the 1-GPU clip) and with the plans of a 16-frame batch
    (policy "shard")."""
    from dsml_thesis_amd.synth import make_tf_model
    from dsml_thesis_amd.ddim import DDIMSampler, C12, C34
    from dsml_thesis_amd.parallel import sample_sharded
    T, W_ = frames, window
    m = make_tf_model(gain=0.25, seq_len=2 * W_ + 1, device=dev)
    rs = np.random.RandomState(2)
    audio = torch.from_numpy(rs.standard_normal((T, 768)).astype(np.float32)).to(dev)
    masked = torch.tanh(torch.from_numpy(rs.standard_normal((T, 3, 128, 128)).astype(np.float32))).to(dev)
    ident = torch.tanh(torch.from_numpy(rs.standard_normal((1, 3, 128, 128)).astype(np.float32))).to(dev)
    c1 = m.cond_stage_model_1.embedding(torch.tensor([[4]], device=dev))
    s = DDIMSampler(m)
    idx = torch.tensor([[min(max(f + i, 0), T - 1) for i in range(-W_, W_ + 1)] for f in range(T)], device=dev)
    xid = m.encode_first_stage(ident)
    c2 = m.cond_stage_model_2(audio[idx])
    cond = {C12: torch.cat([c1.expand(T, -1, -1), c2], dim=2), C34: torch.cat([m.encode_first_stage(masked), xid.expand(T, -1, -1, -1)], dim=1)}
    res = {}
    for pol, items in (("job", job_frames), ("shard", frames)):
        fn = lambda: sample_sharded(s, ddim_steps, T, (3, 32, 32), lambda lo, hi: {k: v[lo:hi] for k, v in cond.items()}, seed=5,
                                    _policy_items=items)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        res[pol] = round(time.perf_counter() - t0, 4)
        assert torch.isfinite(out).all()
    return {"workload": f"{frames} frames (one rank's block of the {job_frames}-frame clip on 8 GPUs), DDIM-{ddim_steps} + decode, one GPU",
            "seconds_plans_of_the_job": res["job"], "seconds_plans_of_the_shard": res["shard"]}


def batch1_leg(dev, frames=4, ddim_steps=50, window=8):
    """The mode the reference SHIPS for talking faces (talking_face/progressive_sampling_difftalk.py:282-317, batch size 1 at
    :350): frames are a serial chain -- the identity latent of frame k+1 is the latent generated for frame k -- so every DDIM
    step is one UNet evaluation at batch 1: a chain of dependent launches, not a throughput problem.  Timed: `frames` frames x
    `ddim_steps` hipGraph-replayed steps through DDIMSampler.progressive_sampling(fixed_identity=False), conditioning
    (audio window encoder, first-stage encode) included, after one untimed clip that builds the program and the graph."""
    from dsml_thesis_amd.synth import make_tf_model
    from dsml_thesis_amd.ddim import DDIMSampler
    T, W_ = frames, window
    m = make_tf_model(gain=0.25, seq_len=2 * W_ + 1, device=dev)
    rs = np.random.RandomState(3)
    audio = torch.from_numpy(rs.standard_normal((T, 768)).astype(np.float32)).to(dev)
    masked = torch.tanh(torch.from_numpy(rs.standard_normal((T, 3, 128, 128)).astype(np.float32))).to(dev)
    masked[:, :, 70:, :] = -1.0
    ident = torch.tanh(torch.from_numpy(rs.standard_normal((1, 3, 128, 128)).astype(np.float32))).to(dev)
    x_T = torch.from_numpy(rs.standard_normal((T, 1, 3, 32, 32)).astype(np.float32)).to(dev)
    c1 = m.cond_stage_model_1.embedding(torch.tensor([[4]], device=dev))
    s = DDIMSampler(m)

    def clip():
        xid = m.encode_first_stage(ident)
        fr, _ = s.progressive_sampling(c1, xid, masked, audio, ddim_steps, 1, T, (3, 32, 32), W_, eta=0.0, x_T=x_T,
                                       fixed_identity=False, use_graph=True)
        return fr
    clip()
    torch.cuda.synchronize()
    els = []
    for _ in range(3):
        t0 = time.perf_counter()
        fr = clip()
        torch.cuda.synchronize()
        els.append(time.perf_counter() - t0)
    el = sorted(els)[1]
    assert len(fr) == T and all(torch.isfinite(f).all() for f in fr)
    pg = m.model.diffusion_model.program(1, 32, 32, 1, 6)
    # the reference's real job is a LOOP over 150 test videos (progressive_sampling_difftalk.py:336): V independent clips advanced
    # frame by frame together (progressive_sampling(..., clips=V)) run the same serial chains at the batched rate
    V = 16
    audio_v = [torch.from_numpy(np.random.RandomState(30 + v).standard_normal((T, 768)).astype(np.float32)).to(dev) for v in range(V)]
    masked_v = [masked for _ in range(V)]
    ident_v = torch.tanh(torch.from_numpy(np.random.RandomState(50).standard_normal((V, 3, 128, 128)).astype(np.float32))).to(dev)
    xT_v = [torch.from_numpy(np.random.RandomState(60 + v).standard_normal((T, 1, 3, 32, 32)).astype(np.float32)).to(dev) for v in range(V)]
    c1_v = m.cond_stage_model_1.embedding((torch.arange(V, device=dev) % 8)[:, None])

    def clips():
        xid = torch.cat([m.encode_first_stage(ident_v[v:v + 1]) for v in range(V)])
        fr, _ = s.progressive_sampling(c1_v, xid, masked_v, audio_v, ddim_steps, 1, None, (3, 32, 32), W_, eta=0.0, x_T=xT_v,
                                       clips=V, use_graph=True)
        return fr
    clips()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frv = clips()
    torch.cuda.synchronize()
    elv = time.perf_counter() - t0
    assert len(frv) == V and all(len(f) == T for f in frv) and all(torch.isfinite(x).all() for f in frv for x in f)
    clips16 = {"workload": f"{V} independent autoregressive clips in lock step (progressive_sampling(clips={V})): every frame of every clip is "
                           f"its own serial chain, one batch of {V} per DDIM step; {T} frames x DDIM-{ddim_steps} each",
               "clips": V, "seconds": round(elv, 4), "sample_steps_per_s": round(V * T * ddim_steps / elv, 1),
               "ms_per_step": round(1e3 * elv / (T * ddim_steps), 4),
               "frames_per_s_at_ddim200": round(V / (200 * elv / (T * ddim_steps)), 3)}
    return {"clips16": clips16,
            "workload": f"talking_face autoregressive clip (the reference's shipped mode), batch 1, {T} frames x DDIM-{ddim_steps}, "
                        f"32x32x3 latent, hipGraph step; conditioning encoders included, decode not",
            "frames": T, "ddim_steps": ddim_steps, "seconds": round(el, 4), "ms_per_step": round(1e3 * el / (T * ddim_steps), 4),
            "sample_steps_per_s": round(T * ddim_steps / el, 1),
            "frames_per_s_at_ddim200": round(1.0 / (200 * el / (T * ddim_steps)), 3),
            "launches_per_unet_eval": len(pg.calls),
            "route": "small-batch program (unet_small.py)" if getattr(pg, "small_route", False) else "batched program"}


def end_to_end_leg(dev, latent, batch, ddim_steps=200):
    """BASELINE configs[1] END TO END, the native counterpart of face_reenactment/sample_affectnet.py:66-137 (what
    tools/sample_faces.py prints): inside ema_scope, DDIMSampler.sample (hipGraph step) at classifier-free-guidance scale 1.0
    and 3.0 (batch doubled: two UNet evaluations per sample-step), then decode_first_stage + clamp to [0,1] NHWC.  Each setting
    runs twice; the second run (programs built, graph captured) is timed."""
    from dsml_thesis_amd.ddim import DDIMSampler
    from dsml_thesis_amd import ops
    model, ucfg = build_model(latent, dev)
    sampler = DDIMSampler(model)
    labels = (torch.arange(batch, device=dev) % 8)[:, None]
    res = {}
    with model.ema_scope():
        c = model.cond_stage_model.embedding(labels)
        for scale in (1.0, 3.0):
            uc = model.cond_stage_model.uncond_embedding(torch.zeros_like(labels)) if scale > 1.0 else None
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                z, _ = sampler.sample(S=ddim_steps, batch_size=batch, shape=[ucfg["in_channels"], latent, latent], conditioning=c, eta=0.0,
                                      unconditional_guidance_scale=scale, unconditional_conditioning=uc, verbose=False, use_graph=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                x = ops.postprocess_frames(model.decode_first_stage(z))
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            assert torch.isfinite(x).all()
            res[f"cfg{scale:g}"] = {"seconds": round(t2 - t0, 3), "frames_per_s": round(batch / (t2 - t0), 3),
                                    "sampling_seconds": round(t1 - t0, 3), "sample_steps_per_s": round(batch * ddim_steps / (t1 - t0), 1),
                                    "unet_evals_per_sample_step": 2 if scale > 1.0 else 1, "decode_seconds": round(t2 - t1, 4),
                                    "decode_frames_per_s": round(batch / (t2 - t1), 1), "shape": list(x.shape)}
    res["workload"] = (f"class-conditional faces, {batch} samples, DDIM-{ddim_steps}, {latent}x{latent}x{ucfg['in_channels']} latent -> "
                       f"{x.shape[1]}x{x.shape[2]} frames, ema_scope, sample() + decode_first_stage + clamp")
    # BASELINE configs[0] on the GPU: DDIM 50 steps, batch 1, this latent, every call inside its own ema_scope (the per-call fixed
    # costs -- schedule, scope enter / exit, flag read -- weigh most here); the model's null-class token as conditioning
    lab1 = torch.zeros(1, 1, dtype=torch.long, device=dev)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with model.ema_scope():
            uc1 = model.cond_stage_model.uncond_embedding(lab1)
            z1, _ = sampler.sample(S=50, batch_size=1, shape=[ucfg["in_channels"], latent, latent], conditioning=uc1, eta=0.0, verbose=False,
                                   use_graph=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    assert torch.isfinite(z1).all()
    res["config0_ddim50_batch1"] = {"seconds": round(min(ts[1:]), 4), "first_call_seconds": round(ts[0], 4), "ms_per_step": round(1e3 * min(ts[1:]) / 50, 3),
                                    "workload": f"`with ema_scope(): sample(S=50, batch_size=1)` at {latent}x{latent}x{ucfg['in_channels']}, scope + schedule + 50 graph-replayed steps"}
    return res


def train_mode(a, rank, world, dev, dist, backend, barrier, graph, emit):
    """`bench.py --train`: BASELINE configs[4] as the one JSON line."""
    latent = 32 if a.latent == 64 and "--latent" not in " ".join(sys.argv) else a.latent
    out = train_measure(a, latent, rank, world, dev, dist, backend, barrier, graph)
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_measure(a, latent, rank, world, dev, dist, backend, barrier, graph):
    """BASELINE configs[4]: one optimisation step = q_sample + UNet forward + hand-written backward
    (hipGraph-captured) + gradient all-reduce over the ranks + AdamW + EMA, fixed batch per GPU (weak scaling).
    `a`: anything with .batch / .bf16 / .steps / .warmup."""
    from dsml_thesis_amd.train import UNetTrainer
    model, ucfg = build_model(latent, dev)
    tr = UNetTrainer(model.model.diffusion_model, compute="bf16" if a.bf16 else "f32")
    sa, sb = model.sqrt_alphas_cumprod, model.sqrt_one_minus_alphas_cumprod
    n, c = a.batch, ucfg["in_channels"]
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    x0 = torch.randn(n, c, latent, latent, generator=g).to(dev)
    noise = torch.randn(n, ucfg["out_channels"], latent, latent, generator=g).to(dev)
    ctx = torch.randn(n, 1, ucfg["context_dim"], generator=g).to(dev)
    t = torch.randint(0, 1000, (n,), generator=g).to(dev)
    shadow = tr.P.flat.clone()
    loss_buf = torch.zeros(1, device=dev)

    def fwd_bwd():
        # N > 1: eager launches (the host runs ahead of the GPU) with the gradient buckets all-reduced while the
        # backward is still running; N = 1: the whole forward+backward is one hipGraph
        loss_buf.copy_(tr.p_losses(x0, ctx, t, noise, sa, sb, reduce_world=world))

    for _ in range(2):
        fwd_bwd()
    torch.cuda.synchronize()
    run = fwd_bwd
    graph = graph and dist is None
    if graph:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fwd_bwd()
        run = gr.replay

    def step():
        run()
        if dist is not None and world == 1:       # LDMK_BENCH_FORCE_DIST rehearsal: the collective in the form backward() issues it
            w = dist.all_reduce(tr.P.grad[tr.P.grad.numel() // 2:], async_op=True)
            w.wait()
        tr.adamw_step(lr=1e-6)
        tr.ema_update(shadow, 0.9999)

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = tt.item()
    assert torch.isfinite(loss_buf).all()
    fwd_gemm = (42.17 if latent == 32 else 168.62) * 1e9 * n
    fwd_attn = (3.84 if latent == 32 else 61.43) * 1e9 * n
    flops = world * (3 * fwd_gemm + 3.5 * fwd_attn)
    ms = 1e3 * el / a.steps
    out = {"metric": "UNet training samples/sec (p_losses forward+backward+AdamW+EMA), BASELINE configs[4]",
           "value": round(world * n * a.steps / el, 2), "unit": "samples/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16" if a.bf16 else "f32", "data": "synthetic",
           "config": {"workload": f"face_reenactment UNet fine-tune step (latent_manipulation_tuned.py / main.py -t), "
                                  f"{n} samples/GPU, {latent}x{latent}x{c} latent, "
                                  + ("bf16 matrix-core GEMMs and attention products (fp32 accumulate, master weights, norms, softmax)" if a.bf16
                                     else "fp32 (the parity path; --bf16 selects BASELINE's bf16 compute)")
                                  + ", AdamW + EMA, random-init weights", "batch_per_gpu": n,
                      "global_batch": n * world, "hipgraph": graph,
                      "parallelism": f"dp{world} (flat gradient buffer all-reduced in 128 MB buckets, overlapped with the backward)"},
           "step_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
           # priced against the peak of the arithmetic the products run in: bf16 matrix cores for --bf16 (the step is then far
           # from that roofline -- its GEMMs stage fp32 activations: DESIGN section 9), the f32 matrix rate otherwise
           "roofline": {"bound": "mfma", "achieved": round(flops / world / (ms * 1e-3) / 1e12, 2),
                        "peak": PEAK_BF16_MFMA if a.bf16 else PEAK_F32_MFMA, "unit": "TFLOP/s",
                        "frac": round(flops / world / (ms * 1e-3) / 1e12 / (PEAK_BF16_MFMA if a.bf16 else PEAK_F32_MFMA), 4),
                        "traffic": None, "kernel": "whole step (igemm + wgrad + attention fwd/bwd), 3x forward GEMM FLOPs"},
           "cpu_baseline": None, "loss": float(loss_buf.item())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="samples per GPU (BASELINE configs[1]: 16)")
    ap.add_argument("--latent", type=int, default=64, choices=[32, 64])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="skip the talking-face clip leg (BASELINE configs[2]/[3])")
    ap.add_argument("--no-extras", action="store_true", help="skip the end-to-end (configs[1] sample + decode) and train_bf16 (configs[4]) legs")
    ap.add_argument("--clip-steps", type=int, default=None,
                    help="DDIM steps of the clip leg; default: the shipped 200 (talking_face/sample.sh:27)")
    ap.add_argument("--clip-frames", type=int, default=128)
    ap.add_argument("--clip-policy", default="shard", choices=["shard", "job"],
                    help="launch plans of the sharded clip: tuned for the rank's own frames (shard) or pinned to the whole clip (job: "
                         "bitwise rank-count independence)")
    ap.add_argument("--bf16", action="store_true", help="with --train: bf16 matrix-core compute for every GEMM of the step "
                    "(BASELINE configs[4]); fp32 master weights, accumulation, normalisations and attention")
    ap.add_argument("--train", action="store_true",
                    help="measure BASELINE configs[4] instead (UNet p_losses forward+backward+AdamW+EMA, fp32, data-"
                         "parallel with one all-reduce of the flat gradient buffer); not the default metric")
    a = ap.parse_args()
    if a.clip_steps is None:
        a.clip_steps = 200

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: start the ranks ourselves.  This process has not initialised the GPU (importing torch
        # does not), it only waits for the child and relays its output -- no exec of a GPU-holding process.
        import subprocess
        # the launcher picks the rendezvous port itself (c10d store on 127.0.0.1:0): a port chosen here, released and passed
        # on could be taken in between -- the bind-then-release race the tests got rid of with their file rendezvous
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", "--local-addr=127.0.0.1",
               os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    # stdout carries exactly ONE line (the JSON): everything else this process or its libraries print there -- RCCL's
    # version banner at communicator creation, for one -- goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started {world} ranks")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("LDMK_BENCH_BACKEND", "nccl")      # "gloo": rehearse the N>1 path on a 1-GPU box
    if backend == "gloo":
        local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("LDMK_BENCH_FORCE_DIST"):    # FORCE_DIST: rehearse the RCCL calls with a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("LDMK_BENCH_FORCE_DIST"):
            os.environ["LDMK_FORCE_COLLECTIVE"] = "1"          # parallel.all_gather_items: issue the real all-gather at world 1 too
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    from dsml_thesis_amd.build import build_lib
    if rank == 0:
        build_lib(verbose=False)
    if dist is not None:
        dist.barrier()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(latent, steps, warmup, graph):
        model, ucfg = build_model(latent, dev)
        run = StepRunner(model, ucfg, a.batch, graph=graph, seed=1 + rank)
        for _ in range(warmup):
            run.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run.step()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        assert torch.isfinite(run.pg.inputs["x"]).all(), "non-finite latent after the timed steps"
        # the timed steps ran in the arithmetic the line names: no F16X2 range flag and no folded-LayerNorm guard went up
        # during them, and no site had been moved to another arithmetic (a raised flag would mean saturated, i.e. wrong, products)
        ast = model.model.diffusion_model.arithmetic_status()
        assert not ast["flags_up"] and not ast["ln_flag_up"] and not ast["denied"], f"arithmetic flags after the timed steps: {ast}"
        run.arithmetic_status = ast
        return run, el

    graph = not a.no_graph
    if a.train:
        train_mode(a, rank, world, dev, dist, backend, barrier, graph, emit)
        return
    run, el = measure(a.latent, a.steps, a.warmup, graph)
    value = world * a.batch * a.steps / el
    ms = 1e3 * el / a.steps
    try:
        base_metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        base_metric = "DDIM denoise steps/sec (64\u00d764\u00d74 latent, 256\u00b2 face) at 1/2/4/8 GPUs"
    out = {
        "metric": base_metric if a.latent == 64 else
                  "DDIM denoise steps/sec (32\u00d732\u00d73 latent, 128\u00b2 face) at 1/2/4/8 GPUs",
        "value": round(value, 2), "unit": "sample-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        # every tensor, accumulation, norm and softmax is fp32; with LDMK_SPLIT_BF16 on (default) the large GEMMs and the self
        # attention form their fp32 products from exact three-way bf16 splits on the bf16 matrix cores (include/ldmk.h,
        # LDMK_COMPUTE_BF16X3): same accuracy class as the f32 MFMA form (tests/test_split_gpu.py), same parity bounds
        "arithmetic": ("fp32 storage/accumulate; matrix products: f16x2 split (3 fp16 MFMAs per product, fp32 accuracy class; LDMK_F16X2=0: bf16x3 exact split, 6 bf16 MFMAs) where the x3 plan "
                       "table lists the shape, f32 MFMA elsewhere") if os.environ.get("LDMK_SPLIT_BF16", "1") != "0" else
                      "fp32 throughout (f32 MFMA)",
        "config": {"workload": f"face_reenactment emotion-conditioned LDM (AffectNet config), DDIM-200 schedule, "
                               f"{a.batch} samples/GPU, {a.latent}x{a.latent}x{run.x_T.shape[1]} latent, CFG off, "
                               f"eta 0, random-init weights", "batch_per_gpu": a.batch, "global_batch": a.batch * world,
                   "latent": [int(run.x_T.shape[1]), a.latent, a.latent], "ddim_steps": 200,
                   "hipgraph": graph, "parallelism": f"dp{world} (independent samples per rank, no data-path collective; the "
                                                   f"sharded-clip leg with its all-gather is reported under 'clip')"},
        "arithmetic_flags": {"f16x2_sites": run.arithmetic_status["sites"], "sites_denied": len(run.arithmetic_status["denied"]),
                             "range_flags_up_after_timed_region": len(run.arithmetic_status["flags_up"]),
                             "layernorm_guard_up": run.arithmetic_status["ln_flag_up"]},
        "batch_steps_per_s": round(world * a.steps / el, 3),
        "step_tflops": round(world * GFLOP_STEP[a.latent] * a.batch * 1e-3 / (ms * 1e-3), 2),
    }
    if rank == 0:
        t_ig, n_ig = run.igemm_time_per_step()
        fl_exec = executed_gemm_flops(run.pg) * 1e-12             # TFLOP the GEMM launches of one step execute
        ach = fl_exec / (t_ig * 1e-3)
        fl_alg = algorithmic_gemm_flops(run.pg) * 1e-12           # the same layers in the reference's arithmetic
        t_tr = getattr(run, "winograd_transform_ms", 0.0)
        traffic, tnote, tfam = None, None, None
        for tname in ("traffic_r05.json", "traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath):
                try:
                    trec = json.load(open(tpath)).get(f"latent{a.latent}_b{a.batch}", {})
                except Exception:
                    trec = {}
                if trec.get("whole_step_bytes") is not None:       # round 5: every kernel family of the step (tools/pmc_traffic.py)
                    traffic = trec["whole_step_bytes"]
                    tfam = {k: round(v["bytes_per_step"]) for k, v in trec["families"].items()}
                    tnote = (f"bytes per step over EVERY launch of the step, by kernel family in `traffic_by_family`; rocprofv3 --pmc "
                             f"FETCH_SIZE(x2)+WRITE_SIZE passes of this command (profiles/{tname}); L2<->fabric, Infinity-Cache hits included")
                    break
                traffic = trec.get("hbm_bytes_per_step")
                if traffic is not None:
                    tnote = (f"bytes per step over the GEMM family only, rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE passes of this "
                             f"command (profiles/{tname}); L2<->fabric, Infinity-Cache hits included")
                    break
        t_x3, n_x3 = getattr(run, "x3_ms", 0.0), getattr(run, "x3_launches", 0)
        t_h2, n_h2 = getattr(run, "h2_ms", 0.0), getattr(run, "h2_launches", 0)
        fl_x3 = executed_gemm_flops(run.pg, 2) * 1e-12
        fl_h2 = executed_gemm_flops(run.pg, 3) * 1e-12
        split = None
        if n_x3 or n_h2:
            # the dominant kernel is a split-arithmetic igemm: price IT against the 16-bit matrix peak on the MFMA FLOPs it issues
            # (bf16x3: 6 per fp32-equivalent FLOP, f16x2: 3); the f32-MFMA launches that remain are reported beside it against
            # their own peak
            t_f, fl_f = t_ig - t_x3 - t_h2, fl_exec - fl_x3 - fl_h2
            split = {"f32_mfma": {"launches": n_ig - n_x3 - n_h2, "ms_per_step": round(t_f, 4), "tflops": round(fl_f / (t_f * 1e-3), 2) if t_f > 0 else None,
                                  "peak": PEAK_F32_MFMA, "frac": round(fl_f / (t_f * 1e-3) / PEAK_F32_MFMA, 4) if t_f > 0 else None},
                     "family_fp32_equivalent_tflops": round(ach, 2)}
            for key, mult, n_, t_, fl_ in (("bf16x3", 6.0, n_x3, t_x3, fl_x3), ("f16x2", 3.0, n_h2, t_h2, fl_h2)):
                if n_:
                    iss = mult * fl_ / (t_ * 1e-3)
                    split[key] = {"launches": n_, "ms_per_step": round(t_, 4), "fp32_equivalent_tflops": round(fl_ / (t_ * 1e-3), 2),
                                  "mfma_instructions_per_product": int(mult), "mfma_tflops_issued": round(iss, 2), "peak": PEAK_BF16_MFMA,
                                  "frac": round(iss / PEAK_BF16_MFMA, 4)}
        # ONE fixed basis per arithmetic setting (never chosen from measured times): with the bf16x3 split arithmetic on (default),
        # `achieved` = the bf16 MFMA FLOPs the bf16x3 launches ISSUE (6 x 2MNK) over their summed duration against the dense bf16
        # matrix peak; with LDMK_SPLIT_BF16=0, executed fp32 FLOPs of the whole family against the f32 matrix peak
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F32_MFMA, 4), "basis": "f32_mfma_executed", "traffic": traffic, "traffic_note": tnote,
                           "traffic_by_family": tfam,
                           # the step's algorithmic bytes (fp32 weights once per step + in/out of every block per sample)
                           "algorithmic_bytes_per_step": ALGO_BYTES[a.latent](a.batch),
                           "traffic_over_algorithmic": round(traffic / ALGO_BYTES[a.latent](a.batch), 2) if traffic else None,
                           "kernel": "ldmk::igemm_kernel<...> + ldmk::rgemm_kernel<...> (every Conv2d / Linear launch of the step)",
                           "flops_basis": "executed: sum of 2*M*N*K (x batch) over the step's GEMM launches -- what the matrix "
                                          "cores do.  The wide 3x3 convolutions run through Winograd F(2x2,3x3), which executes 4/9 "
                                          "of their multiplications: see 'reference_arithmetic' for the same launches in the reference's "
                                          "arithmetic, with the transform kernels' time included",
                           # the same LAUNCHED layers counted in the reference's arithmetic (a direct 3x3 convolution where the
                           # step runs Winograd / phase GEMMs), over GEMM + transform time: an effective rate of work the matrix
                           # cores did NOT all do -- deliberately without a peak fraction; `frac` above is the only roofline value
                           "reference_arithmetic": {"gflop_per_sample_step": round(fl_alg * 1e3 / a.batch, 2),
                                                    "ms_per_step_gemm_plus_winograd_transforms": round(t_ig + t_tr, 4),
                                                    "winograd_transform_ms_per_step": round(t_tr, 4),
                                                    "effective_tflops_reference_basis": round(fl_alg / ((t_ig + t_tr) * 1e-3), 2)},
                           "executed_gflop_per_sample_step": round(fl_exec * 1e3 / a.batch, 2),
                           "reference_gflop_per_sample_step": GFLOP_IGEMM[a.latent],     # a count, no rate is formed on it
                           "launches_per_step": n_ig, "avg_launch_us": round(1e3 * t_ig / n_ig, 2),
                           "sum_launch_ms_per_step": round(t_ig, 4)}
        if split is not None:
            r = out["roofline"]
            dom = "f16x2" if t_h2 >= t_x3 else "bf16x3"
            r["achieved"], r["peak"], r["frac"] = split[dom]["mfma_tflops_issued"], PEAK_BF16_MFMA, split[dom]["frac"]
            r["basis"] = "f16_mfma_issued" if dom == "f16x2" else "bf16_mfma_issued"
            r["fp32_equivalent_tflops"] = split[dom]["fp32_equivalent_tflops"]
            r["kernel"] = ("ldmk::igemm_kernel<..., BF = 4> (LDMK_COMPUTE_F16X2: fp32-accurate products from THREE fp16 MFMAs, include/ldmk.h) "
                           "-- the launches that hold most of the GEMM time; `achieved` counts the fp16 MFMA FLOPs they issue (3 x 2MNK) "
                           "against the nominal dense 16-bit matrix peak.  Fewer issued FLOPs per product is the point of this arithmetic: "
                           "`fp32_equivalent_tflops` is the rate of useful work, `sustained` what the chip holds on this instruction"
                           if dom == "f16x2" else
                           "ldmk::igemm_kernel<..., BF = 3> / igemm_ps_kernel / igemm_pw_kernel (LDMK_COMPUTE_BF16X3: fp32-accurate "
                           "products from six bf16 MFMAs, include/ldmk.h; the ps / pw forms take both operands pre-split and stage them by "
                           "LDS-DMA) -- the launches that hold most of the GEMM time; `achieved` counts the bf16 MFMA FLOPs they issue "
                           "(6 x 2MNK) against the dense bf16 matrix peak")
            # what the chip SUSTAINS on back-to-back 16-bit MFMAs from registers with real operands (power-limited: tools/clock_probe.hip,
            # measured on another box of this pool and committed; the nominal peak assumes 2.4 GHz)
            sus = sustained_mfma16_tflops()
            if sus is not None:
                r["sustained"] = {"mfma16_tflops_register_loop": sus, "source": "profiles/r04_clock_bf16.txt (tools/clock_probe.hip, median of the "
                                  "bf16 runs on pseudo-random operands)", "frac_of_sustained": round(r["achieved"] / sus, 4)}
            r["by_arithmetic"] = split
    del run
    torch.cuda.empty_cache()
    if not a.no_secondary and world == 1:
        other = 32 if a.latent == 64 else 64
        run2, el2 = measure(other, max(5, a.steps // 2), 2, graph)
        out["secondary"] = {"workload": f"same, {other}x{other}x{run2.x_T.shape[1]} latent",
                            "value": round(a.batch * max(5, a.steps // 2) / el2, 2), "unit": "sample-steps/s",
                            "ms_per_step": round(1e3 * el2 / max(5, a.steps // 2), 4)}
        del run2
        if os.environ.get("LDMK_SPLIT_BF16", "1") != "0":
            # the same step with EVERY product on the f32 matrix cores (the arithmetic of rounds 1-2), measured in this very process
            # on this very box, so that `value` (bf16x3 split products, fp32-accurate: DESIGN section 11) has its f32-MFMA twin next to it
            from dsml_thesis_amd import engine as _eng
            torch.cuda.empty_cache()
            os.environ["LDMK_SPLIT_BF16"] = "0"
            _eng.reset_tables()
            try:
                n3 = max(5, a.steps // 2)
                run3, el3 = measure(a.latent, n3, 2, graph)
                out["f32_mfma_form"] = {"workload": "the primary workload with LDMK_SPLIT_BF16=0 (every GEMM and the attention on v_mfma_f32_32x32x2_f32)",
                                        "value": round(a.batch * n3 / el3, 2), "unit": "sample-steps/s", "ms_per_step": round(1e3 * el3 / n3, 4)}
                del run3
            finally:
                del os.environ["LDMK_SPLIT_BF16"]
                _eng.reset_tables()
            if os.environ.get("LDMK_F16X2", "1") != "0":
                # ... and its OPERAND-EXACT twin on the same kernels: every split product from the exact three-way bf16 split (six
                # bf16 MFMAs, no operand perturbed) instead of F16X2's three fp16 ones -- what a site runs after a range-flag fall-back
                torch.cuda.empty_cache()
                os.environ["LDMK_F16X2"] = "0"
                _eng.reset_tables()
                try:
                    n4 = max(5, a.steps // 2)
                    run4, el4 = measure(a.latent, n4, 2, graph)
                    out["bf16x3_form"] = {"workload": "the primary workload with LDMK_F16X2=0 (every split product from the exact bf16x3 split: six bf16 MFMAs, operands bit-exact)",
                                          "value": round(a.batch * n4 / el4, 2), "unit": "sample-steps/s", "ms_per_step": round(1e3 * el4 / n4, 4)}
                    del run4
                finally:
                    del os.environ["LDMK_F16X2"]
                    _eng.reset_tables()
    if not a.no_clip:
        torch.cuda.empty_cache()
        out["clip"] = clip_leg(rank, world, dev, dist, barrier, frames=a.clip_frames, ddim_steps=a.clip_steps, policy=a.clip_policy)
        if rank == 0 and world == 1 and a.clip_frames >= 64 and not a.no_extras:
            # what bitwise rank-count independence costs a shard: ONE GPU's 16-frame block of the 8-GPU job with the plans of
            # the whole 128-frame clip ("job") against the plans of its own batch ("shard")
            from dsml_thesis_amd.parallel import sample_sharded as _ss   # noqa: F401  (documented in clip_leg)
            out["clip"]["shard16_of_128"] = shard_cost_leg(dev, a.clip_steps)
        if rank == 0 and world == 1:
            torch.cuda.empty_cache()
            out["batch1"] = batch1_leg(dev)
    if rank == 0 and world == 1 and not a.no_extras:
        # BASELINE configs[1] end to end and configs[4] (one GPU: forward + backward + AdamW + EMA, bf16 matrix-core compute)
        torch.cuda.empty_cache()
        out["end_to_end"] = end_to_end_leg(dev, a.latent, a.batch)
        torch.cuda.empty_cache()

        class _T:
            batch, bf16, steps, warmup = a.batch, True, 3, 1
        tb = train_measure(_T, a.latent, rank, world, dev, None, backend, barrier, graph)
        out["train_bf16"] = {k: tb[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "step_tflops", "roofline", "loss")}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.latent)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
