"""Generate tests/golden/*.npz from the REAL reference (container only) and pin the oracle.

Usage (from /root/repo):   python tools/make_golden.py            # runs both trees
                           python tools/make_golden.py --tree face_reenactment
For every fixture the reference module is instantiated from /root/reference, loaded with the
synthetic weight recipe (oracle/weights.py), run on seeded inputs on CPU, compared with the
oracle restatement (oracle/ldm_oracle.py) and the *reference's* output is stored.  The GPU
box never sees the reference: tests regenerate weights/inputs from seeds and compare with
the stored outputs.
"""
import argparse
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import ldm_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402


def rnd(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def load_recipe(module, seed=0, gain=1.0, prefix_check=None):
    """Load recipe weights into a reference module; returns the state dict used."""
    ref_sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in ref_sd.items() if v.dtype.is_floating_point and v.dim() > 0}
    if prefix_check is not None:
        mine = prefix_check
        assert set(mine.keys()) <= set(ref_sd.keys()), sorted(set(mine.keys()) - set(ref_sd.keys()))[:5]
        for k, s in mine.items():
            assert tuple(ref_sd[k].shape) == tuple(s), (k, ref_sd[k].shape, s)
        missing = [k for k in shapes if k not in mine]
        assert not missing, f"oracle enumeration misses {missing[:5]}"
    sd = W.synth_state_dict(shapes, seed=seed, gain=gain)
    module.load_state_dict(sd, strict=False)
    module.eval()
    return sd


def check(name, ref, mine, rtol=1e-5, atol=1e-5):
    ref = torch.as_tensor(ref)
    mine = torch.as_tensor(mine)
    err = (ref.double() - mine.double()).abs().max().item()
    scale = ref.double().abs().max().item()
    print(f"  {name:38s} max|ref|={scale:10.4g}  max|ref-oracle|={err:9.3g}")
    torch.testing.assert_close(mine, ref, rtol=rtol, atol=atol)


def save(fname, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(GOLD, fname)
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------- FR tree
def gen_fr():
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.modules.diffusionmodules import openaimodel as om
    from ldm.modules.diffusionmodules import util as ru
    from ldm.modules.diffusionmodules import model as rm
    from ldm.modules import attention as ra
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from taming.modules.vqvae.quantize import VectorQuantizer2
    torch.set_grad_enabled(False)

    # ---- G1 schedules ------------------------------------------------------------------
    print("[G1] schedules")
    betas = ru.make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0205)
    sched = O.register_schedule(**W.SCHEDULE)
    g1 = dict(betas=betas.astype(np.float32))
    check("betas", torch.tensor(betas, dtype=torch.float32), sched["betas"], 0, 0)

    # a real LatentDiffusion from the shipped YAML values (plain dicts; no checkpoints offline)
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.FR_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True,
                         conditioning_key="crossattn", unet_config=unet_cfg, image_size=32, channels=3,
                         first_stage_key="image", log_every_t=200, monitor="val_loss_ema", **W.SCHEDULE)
    for name in ("alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
                 "posterior_log_variance_clipped", "sqrt_one_minus_alphas_cumprod"):
        check(name, getattr(ld, name), sched[name], 0, 0)
        g1[name] = getattr(ld, name)

    class CPUDDIM(DDIMSampler):          # the reference hard-codes .to('cuda')
        def register_buffer(self, n, a):
            setattr(self, n, a)

    sampler = CPUDDIM(ld)
    for S in (50, 200):
        for eta in (0.0, 1.0):
            sampler.make_schedule(S, ddim_eta=eta, verbose=False)
            ts = O.make_ddim_timesteps(S)
            assert (ts == sampler.ddim_timesteps).all()
            tab = O.make_ddim_tables(sched["alphas_cumprod"], ts, eta)
            # what torch.full() sees at ddim.py:188-191, squeezed to float32
            ref = dict(
                a_t=np.asarray([torch.full((1,), sampler.ddim_alphas[i]).item() for i in range(S)], np.float32),
                a_prev=np.asarray([torch.full((1,), sampler.ddim_alphas_prev[i]).item() for i in range(S)], np.float32),
                sigma_t=np.asarray([torch.full((1,), sampler.ddim_sigmas[i]).item() for i in range(S)], np.float32),
                sqrt_one_minus_at=np.asarray([torch.full((1,), sampler.ddim_sqrt_one_minus_alphas[i]).item()
                                              for i in range(S)], np.float32))
            for k in ref:
                check(f"ddim S={S} eta={eta} {k}", ref[k], tab[k], 0, 0)
                g1[f"S{S}_eta{int(eta)}_{k}"] = ref[k]
            g1[f"S{S}_timesteps"] = sampler.ddim_timesteps
    save("g1_schedules.npz", **g1)

    # ---- G2 timestep embedding ---------------------------------------------------------
    print("[G2] timestep_embedding")
    t = torch.tensor([0, 1, 500, 999], dtype=torch.long)
    ref = ru.timestep_embedding(t, 160)
    check("timestep_embedding", ref, O.timestep_embedding(t, 160), 0, 0)
    save("g2_timestep_embedding.npz", t=t, emb=ref)

    # ---- G3 per-op ---------------------------------------------------------------------
    print("[G3] per-op")
    g3 = {}
    # GroupNorm32 + SiLU (eps 1e-5), incl. a straddling-concat width (480 = 320+160, 15 ch/group)
    for tag, shape, seed in (("gn160", (2, 160, 8, 8), 11), ("gn480", (1, 480, 4, 4), 12)):
        gn = ru.normalization(shape[1])
        sd = load_recipe(gn, seed=1)
        x = rnd(seed, *shape) * 1.5 + 0.3
        ref = torch.nn.functional.silu(gn(x))
        check(tag, ref, O.gn_silu(x, sd["weight"], sd["bias"]))
        g3[tag] = ref
    # conv3x3 160->320 @8x8
    conv = torch.nn.Conv2d(160, 320, 3, padding=1)
    sd = load_recipe(conv, seed=2)
    x = rnd(13, 1, 160, 8, 8)
    g3["conv3x3"] = conv(x)
    # ResBlock 160->320 @8x8 with emb
    rb = om.ResBlock(160, 640, 0, out_channels=320)
    sd = load_recipe(rb, seed=3)
    x, emb = rnd(14, 2, 160, 8, 8), rnd(15, 2, 640)
    ref = rb(x, emb)
    check("resblock", ref, O.resblock(sd, "", x, emb), 1e-5, 2e-5)
    g3["resblock"] = ref
    # CrossAttention: self (context None) and cross with L in {1, 3}
    ca = ra.CrossAttention(160, context_dim=None, heads=5, dim_head=32)
    sd = load_recipe(ca, seed=4)
    x = rnd(16, 1, 64, 160)
    ref = ca(x)
    check("self-attn", ref, O.cross_attention(sd, "", x, None, 5))
    g3["attn_self"] = ref
    ca = ra.CrossAttention(160, context_dim=512, heads=5, dim_head=32)
    sd = load_recipe(ca, seed=5)
    for L in (1, 3):
        ctx = rnd(17 + L, 1, L, 512)
        ref = ca(x, context=ctx)
        check(f"cross-attn L={L}", ref, O.cross_attention(sd, "", x, ctx, 5))
        g3[f"attn_cross_L{L}"] = ref
    # GEGLU feed-forward
    ff = ra.FeedForward(160, glu=True)
    sd = load_recipe(ff, seed=6)
    ref = ff(x)
    check("geglu-ff", ref, O.geglu_ff(sd, "", x))
    g3["geglu_ff"] = ref
    # SpatialTransformer(160, 5, 32, ctx 512) @8x8
    st = ra.SpatialTransformer(160, 5, 32, depth=1, context_dim=512)
    sd = load_recipe(st, seed=7)
    x4, ctx = rnd(21, 2, 160, 8, 8), rnd(22, 2, 1, 512)
    ref = st(x4, ctx)
    check("spatial-transformer", ref, O.spatial_transformer(sd, "", x4, ctx, 5), 1e-5, 2e-5)
    g3["spatial_transformer"] = ref
    ctx3 = rnd(23, 2, 3, 512)
    ref = st(x4, ctx3)
    check("spatial-transformer L=3", ref, O.spatial_transformer(sd, "", x4, ctx3, 5), 1e-5, 2e-5)
    g3["spatial_transformer_L3"] = ref
    # note: context=None only works when context_dim == inner dim (attention.py:155,174-175); with the
    # shipped context_dim=512/1024 the reference raises a shape error, so there is no such fixture.
    # Downsample / Upsample
    dn = om.Downsample(160, True, dims=2, out_channels=160)
    sd = load_recipe(dn, seed=8)
    g3["downsample"] = dn(x4)
    up = om.Upsample(160, True, dims=2, out_channels=160)
    sd = load_recipe(up, seed=9)
    g3["upsample"] = up(x4)
    # DDIM update at index 100 of S=200, eta=1 with injected noise; DDPM posterior update
    sampler.make_schedule(200, ddim_eta=1.0, verbose=False)
    x, e, nz = rnd(31, 2, 3, 32, 32), rnd(32, 2, 3, 32, 32), rnd(33, 2, 3, 32, 32)
    tab = O.make_ddim_tables(sched["alphas_cumprod"], O.make_ddim_timesteps(200), 1.0)
    idx = 100

    class _M:  # apply_model returns the injected eps
        def apply_model(self, x_, t_, c_):
            return e
    sampler_model = sampler.model
    sampler.model = _M()
    torch.manual_seed(1234)
    xp, px0 = sampler.p_sample_ddim(x, None, torch.full((2,), 501), index=idx)
    sampler.model = sampler_model
    torch.manual_seed(1234)
    nz_ref = torch.randn(x.shape)
    mxp, mpx0 = O.ddim_update(x, e, tab["a_t"][idx], tab["a_prev"][idx], tab["sigma_t"][idx],
                              tab["sqrt_one_minus_at"][idx], nz_ref)
    check("ddim update x_prev", xp, mxp, 1e-6, 1e-6)
    check("ddim update pred_x0", px0, mpx0, 1e-6, 1e-6)
    g3["ddim_x_prev"], g3["ddim_pred_x0"], g3["ddim_noise"] = xp, px0, nz_ref
    t = torch.tensor([0, 700])
    ld_apply = ld.apply_model
    ld.apply_model = lambda x_, t_, c_, return_ids=False: e
    torch.manual_seed(77)
    ref = ld.p_sample(x, None, t, clip_denoised=False)
    ld.apply_model = ld_apply
    torch.manual_seed(77)
    nz_ref = torch.randn(x.shape)
    check("ddpm update", ref, O.ddpm_update(sched, x, e, t, nz_ref), 1e-6, 1e-6)
    g3["ddpm_x_prev"], g3["ddpm_noise"] = ref, nz_ref
    save("g3_ops.npz", **g3)

    # ---- G4 full UNet evals ------------------------------------------------------------
    print("[G4] UNet evals")
    g4 = {}
    unet = ld.model.diffusion_model
    usd = load_recipe(unet, seed=0, prefix_check=W.unet_param_shapes(W.FR_UNET))
    x, t, ctx = rnd(41, 2, 3, 32, 32), torch.tensor([3, 981]), rnd(42, 2, 1, 512)
    ref = unet(x, t, context=ctx)
    check("FR UNet eval", ref, O.unet_forward(usd, W.FR_UNET, x, t, ctx), 1e-4, 1e-4)
    g4["fr_eps"] = ref
    ns = om.UNetModel(**W.NS_UNET)
    nsd = load_recipe(ns, seed=0, prefix_check=W.unet_param_shapes(W.NS_UNET))
    x, t, ctx = rnd(43, 1, 4, 64, 64), torch.tensor([501]), rnd(44, 1, 1, 512)
    ref = ns(x, t, context=ctx)
    check("NS 64x64x4 UNet eval", ref, O.unet_forward(nsd, W.NS_UNET, x, t, ctx), 1e-4, 1e-4)
    g4["ns_eps"] = ref
    del ns, nsd
    save("g4_unet_fr.npz", **g4)

    # ---- G5 short trajectories (gain 0.25 keeps them finite) -----------------
    print("[G5] DDIM / DDPM trajectories")
    g5 = {}
    usd = load_recipe(unet, seed=0, gain=0.25)
    csd = load_recipe(ld.cond_stage_model, seed=0)
    labels = torch.tensor([1, 6])
    c = ld.cond_stage_model.embedding(labels[:, None])
    uc = ld.cond_stage_model.uncond_embedding(torch.zeros(2, 1, dtype=torch.long))
    check("cond embedding", c, csd["embedding.weight"][labels][:, None], 0, 0)
    xT = rnd(51, 2, 3, 32, 32)
    # long runs are too chaotic to commit to as a *tolerance* fixture; pin 3/4-step runs instead

    def ref_ddim(S, nsteps, eta, scale, seed=None):
        sampler.make_schedule(S, ddim_eta=eta, verbose=False)
        img = xT
        tsr = np.flip(sampler.ddim_timesteps)
        if seed is not None:
            torch.manual_seed(seed)
        for i, step in enumerate(tsr[:nsteps]):
            index = S - i - 1
            ts_ = torch.full((2,), int(step), dtype=torch.long)
            img, _ = sampler.p_sample_ddim(img, c, ts_, index=index, unconditional_guidance_scale=scale,
                                           unconditional_conditioning=uc if scale != 1.0 else None)
        return img

    def my_ddim(S, nsteps, eta, scale, seed=None):
        ts = O.make_ddim_timesteps(S)
        tab = O.make_ddim_tables(sched["alphas_cumprod"], ts, eta)
        img = xT
        if seed is not None:
            torch.manual_seed(seed)
        for i, step in enumerate(np.flip(ts)[:nsteps]):
            index = S - i - 1
            t_ = torch.full((2,), int(step), dtype=torch.long)
            if scale == 1.0:
                e = O.apply_model(usd, W.FR_UNET, img, t_, [c])
            else:
                e_u, e_c = O.apply_model(usd, W.FR_UNET, torch.cat([img] * 2), torch.cat([t_] * 2),
                                         [torch.cat([uc, c])]).chunk(2)
                e = O.cfg_combine(e_u, e_c, scale)
            nz = torch.randn(img.shape) if eta > 0 else None
            img, _ = O.ddim_update(img, e, tab["a_t"][index], tab["a_prev"][index], tab["sigma_t"][index],
                                   tab["sqrt_one_minus_at"][index], nz)
        return img

    for tag, (S, n, eta, scale, seed) in dict(s200_e0_cfg1=(200, 3, 0.0, 1.0, None),
                                              s200_e0_cfg3=(200, 3, 0.0, 3.0, None),
                                              s200_e1_cfg1=(200, 3, 1.0, 1.0, 99)).items():
        ref = ref_ddim(S, n, eta, scale, seed)
        check(f"ddim 3-step {tag}", ref, my_ddim(S, n, eta, scale, seed), 1e-4, 1e-4)
        g5[tag] = ref
    # full DDIMSampler.sample with S=4 (every line of ddim_sampling exercised)
    ref, _ = sampler.sample(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False)
    check("DDIMSampler.sample S=4", ref, O.ddim_sample(usd, W.FR_UNET, sched, 4, xT, cond=c), 1e-4, 1e-4)
    g5["sample_S4"] = ref
    # ancestral p_sample_loop, 3 steps with seeded noise
    torch.manual_seed(5)
    ref = ld.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=3, verbose=False)
    torch.manual_seed(5)
    nz = [torch.randn(xT.shape) for _ in range(3)]
    check("p_sample_loop T=3", ref, O.p_sample_loop(usd, W.FR_UNET, sched, xT, cond=c, timesteps=3, noise=nz),
          1e-4, 1e-4)
    g5["p_sample_loop_T3"] = ref
    g5["p_sample_loop_noise"] = torch.stack(nz)
    torch.manual_seed(99)
    g5["s200_e1_noise"] = torch.stack([torch.randn(xT.shape) for _ in range(3)])
    save("g5_sampling_fr.npz", **g5)

    # ---- G8 DDIM inversion + regeneration (latent manipulation) ---------------------
    print("[G8] DDIM inversion (compute_latents.py)")
    import compute_latents as cl          # the driver script: importable once albumentations/OmegaConf are stubbed

    class CPUInv(cl.DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)

    inv = CPUInv(ld)
    x0 = rnd(81, 2, 3, 32, 32)
    g8 = {}
    for tag, scale in (("cfg1", 1.0), ("cfg3", 3.0)):
        img, xlat, _ = inv.compute_latents(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, x0=x0, eta=0.0,
                                           verbose=False, strength=0.5, unconditional_guidance_scale=scale,
                                           unconditional_conditioning=uc if scale != 1.0 else None)
        assert np.array_equal(inv.ddim_timesteps, O.make_ddim_timesteps_strength(4, 1000, 0.5))
        mimg, mlat = O.ddim_invert_and_regenerate(usd, W.FR_UNET, sched, 4, x0, c, strength=0.5, scale=scale,
                                                  uncond=uc if scale != 1.0 else None)
        check(f"inversion latent {tag}", xlat, mlat, 1e-4, 1e-4)
        check(f"regenerated {tag}", img, mimg, 1e-4, 1e-4)
        g8[f"xlat_{tag}"], g8[f"img_{tag}"] = xlat, img
    g8["timesteps"] = inv.ddim_timesteps
    save("g8_inversion.npz", **g8)

    # ---- G6 VQGAN first stage ----------------------------------------------------------
    print("[G6] VQGAN")
    g6 = {}
    fsm = ld.first_stage_model
    vsd = load_recipe(fsm, seed=0, prefix_check=W.vqmodel_param_shapes(W.VQ_F4))
    z = rnd(61, 1, 3, 32, 32)
    zq, _, (_, _, idx) = fsm.quantize(z)
    mzq, midx = O.vq_quantize(z, vsd["quantize.embedding.weight"])
    assert (idx == midx).all()
    check("vq z_q", zq, mzq, 0, 1e-7)
    g6["vq_idx"], g6["vq_zq"] = idx.to(torch.int32), zq
    ab = rm.AttnBlock(512)
    sd = load_recipe(ab, seed=1)
    x = rnd(62, 1, 512, 8, 8)
    ref = ab(x)
    check("AttnBlock", ref, O.vq_attn_block(sd, "", x), 1e-5, 2e-5)
    g6["attn_block"] = ref
    rb = rm.ResnetBlock(in_channels=256, out_channels=128, dropout=0.0, temb_channels=0)
    sd = load_recipe(rb, seed=2)
    x = rnd(63, 1, 256, 8, 8)
    ref = rb(x, None)
    check("ResnetBlock", ref, O.vq_resnet_block(sd, "", x), 1e-5, 2e-5)
    g6["resnet_block"] = ref
    ref = ld.decode_first_stage(z)
    mine, _ = O.decode_first_stage(vsd, W.VQ_F4, z)
    check("decode_first_stage", ref, mine, 1e-4, 2e-4)
    g6["decoded"] = ref.half()
    g6["decoded_stats"] = np.asarray([ref.abs().max().item(), ref.mean().item(), ref.std().item()])
    img = torch.tanh(rnd(64, 1, 3, 128, 128))
    ref = ld.first_stage_model.encode(img)
    mine = O.encode_first_stage(vsd, W.VQ_F4, img)
    check("encode_first_stage", ref, mine, 1e-4, 2e-4)
    g6["encoded"] = ref
    save("g6_vqgan.npz", **g6)


# --------------------------------------------------------------------------- training step (N1)
TRAIN_UNET = dict(W.FR_UNET, model_channels=64, channel_mult=[1, 2], num_res_blocks=1, attention_resolutions=[2, 1])
TRAIN_FULL_KEYS = ("out.2.bias", "time_embed.0.bias", "input_blocks.1.1.norm.weight", "input_blocks.0.0.bias",
                   "middle_block.1.transformer_blocks.0.norm3.bias", "output_blocks.1.0.skip_connection.bias",
                   "input_blocks.2.0.op.bias", "output_blocks.0.2.conv.bias")


def gen_train():
    """G9: the reference's own LatentDiffusion.p_losses (ddpm.py:1014-1047) + autograd backward on a reduced UNet:
    loss, per-parameter gradient (sum, L2 norm), a few gradients in full, and the gradient w.r.t. the context."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.models.diffusion.ddpm import LatentDiffusion
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(TRAIN_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True, conditioning_key="crossattn",
                         unet_config=unet_cfg, image_size=16, channels=3, first_stage_key="image", log_every_t=200,
                         monitor="val_loss_ema", **W.SCHEDULE)
    unet = ld.model.diffusion_model
    load_recipe(unet, seed=0, prefix_check=W.unet_param_shapes(TRAIN_UNET))
    ld.train()
    torch.set_grad_enabled(True)
    n = 2
    x0, noise = rnd(101, n, 3, 16, 16), rnd(102, n, 3, 16, 16)
    c = rnd(103, n, 1, 512).requires_grad_(True)
    t = torch.tensor([17, 803])
    loss, loss_dict = ld.p_losses(x0, c, t, noise=noise)
    loss.backward()
    g9 = dict(loss=loss.detach(), loss_simple=loss_dict["train_loss_simple"].detach(), dcontext=c.grad)
    names, stats = [], []
    for k, p_ in unet.named_parameters():
        g = torch.zeros_like(p_) if p_.grad is None else p_.grad
        names.append(k)
        stats.append([g.double().sum().item(), g.double().norm().item()])
        if k in TRAIN_FULL_KEYS:
            g9["grad:" + k] = g
    g9["names"] = np.asarray(names)
    g9["stats"] = np.asarray(stats, dtype=np.float64)
    save("g9_p_losses.npz", **g9)
    torch.set_grad_enabled(False)


def gen_diffclip():
    """G10 (row N2): the reference's differentiable DDIM (ddim2.DDIMSampler2.differentiable_p_sample_ddim, CFG by batch
    doubling) + differentiable_decode_first_stage (ddpm.py:767-824) + an l2 image loss, autograd backward:
    decoded image, loss, d(loss)/d(x), per-parameter UNet gradient norms."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.ddim2 import DDIMSampler2
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(TRAIN_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True, conditioning_key="crossattn",
                         unet_config=unet_cfg, image_size=16, channels=3, first_stage_key="image", log_every_t=200,
                         monitor="val_loss_ema", **W.SCHEDULE)
    unet = ld.model.diffusion_model
    load_recipe(unet, seed=0, prefix_check=W.unet_param_shapes(TRAIN_UNET))
    load_recipe(ld.first_stage_model, seed=0, prefix_check=W.vqmodel_param_shapes(W.VQ_F4))
    ld.train()

    class CPU2(DDIMSampler2):
        def register_buffer(self, n, a):
            setattr(self, n, a)

    sm = CPU2(ld)
    sm.make_schedule(ddim_num_steps=3, ddim_eta=0.0, strength=0.3, verbose=False)
    torch.set_grad_enabled(True)
    x = rnd(401, 1, 3, 16, 16).requires_grad_(True)
    x0 = torch.tanh(rnd(402, 1, 3, 64, 64))
    c, uc = rnd(403, 1, 1, 512), rnd(404, 1, 1, 512)
    ts = sm.ddim_timesteps
    xi = x
    for i, step in enumerate(np.flip(ts)):
        index = len(ts) - i - 1
        tt = torch.full((1,), int(step), dtype=torch.long)
        xi, _ = sm.differentiable_p_sample_ddim(x=xi, c=c, t=tt, index=index, unconditional_guidance_scale=2.0,
                                                 unconditional_conditioning=uc)
    img = ld.differentiable_decode_first_stage(xi)
    loss = torch.nn.functional.mse_loss(img, x0)
    loss.backward()
    names, stats = [], []
    for k, p_ in unet.named_parameters():
        g = torch.zeros_like(p_) if p_.grad is None else p_.grad
        names.append(k)
        stats.append([g.double().sum().item(), g.double().norm().item()])
    save("g10_diffclip.npz", timesteps=ts, z=xi.detach(), image=img.detach().half(), loss=loss.detach(), dx=x.grad,
         image_stats=np.asarray([img.abs().max().item(), img.mean().item(), img.std().item()]),
         names=np.asarray(names), stats=np.asarray(stats, dtype=np.float64))
    torch.set_grad_enabled(False)


# --------------------------------------------------------------------------- TF tree
def gen_tf():
    from tools import ref_shims
    ref_shims.install("talking_face")
    from ldm.models.diffusion.ddim2cond import DDIMSampler
    from ldm.models.diffusion.ddpm2cond import LatentDiffusion
    torch.set_grad_enabled(False)
    sched = O.register_schedule(**W.SCHEDULE)
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.TF_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    c1_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder",
                  params=dict(embed_dim=256, n_classes=8, key="class_label", p_uncond=0.2))
    W_AUDIO = 1                                  # window 2W+1 = 3 taps for the fixture
    c2_cfg = dict(target="ldm.modules.encoders.modules.Conv1DTemporalAttention",
                  params=dict(seq_len=2 * W_AUDIO + 1, subspace_dim=768, subspace2hidden=False))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config_1=c1_cfg, cond_stage_config_2=c2_cfg,
                         num_timesteps_cond=1, cond_stage_key_1="class_label", cond_stage_key_2="audio",
                         cond_stage_trainable=True, conditioning_key="crossattn", unet_config=unet_cfg,
                         image_size=32, channels=3, first_stage_key="image", log_every_t=200,
                         monitor="val_loss_ema", **W.SCHEDULE)

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)

    g = {}
    unet = ld.model.diffusion_model
    usd = load_recipe(unet, seed=0, prefix_check=W.unet_param_shapes(W.TF_UNET))
    print("[G4-TF] UNet eval (in 9 = x + motion&id concat, ctx 1024)")
    x, t = rnd(71, 2, 3, 32, 32), torch.tensor([11, 756])
    c12, c34 = rnd(72, 2, 1, 1024), rnd(73, 2, 6, 32, 32)
    ref = ld.apply_model(x, t, c12, c34)
    check("TF apply_model", ref, O.apply_model(usd, W.TF_UNET, x, t, [c12], [c34]), 1e-4, 1e-4)
    g["tf_eps"] = ref

    print("[G7] audio attention + progressive sampling (T=3 frames, S=4)")
    asd = load_recipe(ld.cond_stage_model_2, seed=0,
                      prefix_check=W.audio_attention_param_shapes(2 * W_AUDIO + 1))
    a = rnd(74, 2, 3, 768)
    ref = ld.cond_stage_model_2(a)
    check("Conv1DTemporalAttention", ref, O.audio_temporal_attention(asd, a), 1e-5, 1e-5)
    g["audio_att"] = ref
    csd = load_recipe(ld.cond_stage_model_1, seed=0)
    vsd = load_recipe(ld.first_stage_model, seed=0, prefix_check=W.vqmodel_param_shapes(W.VQ_F4))
    usd = load_recipe(unet, seed=0, gain=0.25)
    T, S = 3, 4                                   # S must divide 1000 (util.py:49-50,57 index 1000 otherwise)
    audio = rnd(75, T, 768)
    masked = torch.tanh(rnd(76, T, 3, 128, 128))
    masked[:, :, 70:, :] = -1.0                   # lower face masked to -1 (custom.py:380-387)
    ident = torch.tanh(rnd(77, 1, 3, 128, 128))
    c1 = ld.cond_stage_model_1.embedding(torch.tensor([[4]]))
    xid = ld.encode_first_stage(ident)
    check("encode identity", xid, O.encode_first_stage(vsd, W.VQ_F4, ident), 1e-4, 2e-4)
    g["xid"] = xid
    xT = rnd(78, T, 1, 3, 32, 32)
    sampler = CPUDDIM(ld)
    # the reference's progressive_sampling lives in a driver script that cannot be imported
    # (albumentations/librosa/cv2 at import); drive the *reference's* sampler/model methods with the
    # loop semantics of progressive_sampling_difftalk.py:282-317.
    sampler.make_schedule(S, ddim_eta=0.0, verbose=False)

    def ref_frames(fixed_identity):
        zid = xid.clone()
        frames = []
        for f in range(T):
            idx = [min(max(f + i, 0), T - 1) for i in range(-W_AUDIO, W_AUDIO + 1)]
            c2 = ld.cond_stage_model_2(audio[idx].unsqueeze(0))
            c12_ = torch.cat([c1, c2], dim=2)
            c3 = ld.encode_first_stage(masked[f].unsqueeze(0))
            c34_ = torch.cat([c3, zid], dim=1)
            c = {"class_label_&_audio": c12_, "motion_&_id": c34_}
            img = xT[f]
            for i, step in enumerate(np.flip(sampler.ddim_timesteps)):
                index = S - i - 1
                ts_ = torch.full((1,), int(step), dtype=torch.long)
                img, _ = sampler.p_sample_ddim(x=img, c=c, t=ts_, index=index)
            frames.append(img)
            if not fixed_identity:
                zid = img.clone()
        return torch.cat(frames)

    for fixed in (False, True):
        ref = ref_frames(fixed)
        mine = torch.cat(O.progressive_sampling(usd, W.TF_UNET, sched, vsd, W.VQ_F4, asd, c1, xid, masked,
                                                audio, S, W_AUDIO, xT, fixed_identity=fixed))
        tag = "fixed" if fixed else "autoreg"
        check(f"progressive {tag}", ref, mine, 1e-4, 2e-4)
        g[f"frames_{tag}"] = ref
    # batched fixed-identity == DDIMSampler.sample with batched conditioning (SURVEY F2 mode b)
    c2b = torch.cat([ld.cond_stage_model_2(audio[[min(max(f + i, 0), T - 1) for i in range(-W_AUDIO, W_AUDIO + 1)]]
                                           .unsqueeze(0)) for f in range(T)])
    c12b = torch.cat([c1.expand(T, -1, -1), c2b], dim=2)
    c34b = torch.cat([ld.encode_first_stage(masked), xid.expand(T, -1, -1, -1)], dim=1)
    refb, _ = sampler.sample(S=S, batch_size=T, shape=[3, 32, 32],
                             conditioning={"class_label_&_audio": c12b, "motion_&_id": c34b},
                             eta=0.0, x_T=xT[:, 0], verbose=False)
    check("batched sample == per-frame fixed", refb, g["frames_fixed"], 1e-4, 2e-4)
    g["frames_fixed_batched"] = refb
    save("g7_talking_face.npz", **g)


# --------------------------------------------------------------------------- north-star shapes (BASELINE.json metric)
def gen_northstar():
    """G11: the shapes BASELINE.json quotes its metric on (64x64x4 latent -> 256x256 face), from the real reference:
    a full DDIMSampler.sample trajectory (S=4, B=2), the dim-4 / 16384-code quantiser on 4096 latent vectors (indices
    must match bit for bit), and decode_first_stage 4x64x64 -> 3x256x256 stored in fp32."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    torch.set_grad_enabled(False)
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.NS_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=4, n_embed=16384, ddconfig=dict(W.VQ_F4_256["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True,
                         conditioning_key="crossattn", unet_config=unet_cfg, image_size=64, channels=4,
                         first_stage_key="image", log_every_t=200, monitor="val_loss_ema", **W.SCHEDULE)
    sched = O.register_schedule(**W.SCHEDULE)

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)

    g = {}
    print("[G11] 64x64x4 DDIM trajectory")
    usd = load_recipe(ld.model.diffusion_model, seed=0, gain=0.25, prefix_check=W.unet_param_shapes(W.NS_UNET))
    csd = load_recipe(ld.cond_stage_model, seed=0)
    labels = torch.tensor([3, 4])
    c = ld.cond_stage_model.embedding(labels[:, None])
    xT = rnd(111, 2, 4, 64, 64)
    ref, inter = CPUDDIM(ld).sample(S=4, batch_size=2, shape=[4, 64, 64], conditioning=c, eta=0.0, x_T=xT, verbose=False,
                                    log_every_t=1)
    mine = O.ddim_sample(usd, W.NS_UNET, sched, 4, xT, cond=csd["embedding.weight"][labels][:, None])
    check("DDIMSampler.sample S=4 @64x64x4", ref, mine, 1e-4, 1e-4)
    g["sample_S4"] = ref
    g["x_inter_1"] = inter["x_inter"][1]          # after the first step: localises a failure

    print("[G11] dim-4 / 16384-code quantiser")
    fsm = ld.first_stage_model
    vsd = load_recipe(fsm, seed=0, prefix_check=W.vqmodel_param_shapes(W.VQ_F4_256))
    z = rnd(112, 1, 4, 64, 64)
    zq, _, (_, _, idx) = fsm.quantize(z)
    mzq, midx = O.vq_quantize(z, vsd["quantize.embedding.weight"])
    assert (idx == midx).all()
    check("vq4 z_q", zq, mzq, 0, 0)
    g["vq4_idx"], g["vq4_zq"] = idx.to(torch.int32), zq
    # the rounding sequence of the distance (what the HIP kernel restates): squares rounded then added left to right,
    # k-ordered fma dot product, fl(fl(zz + ee) - 2 ze); checked here on all 4096 x 16384 distances
    zf = z.permute(0, 2, 3, 1).reshape(-1, 4)
    e = vsd["quantize.embedding.weight"]
    d_ref = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(e ** 2, dim=1) - 2 * torch.einsum("bd,dn->bn", zf, e.t())

    def sq(x):
        acc = x[:, 0] * x[:, 0]
        for k in range(1, x.shape[1]):
            acc = acc + x[:, k] * x[:, k]
        return acc

    ze = (zf[:, :1].double() * e[None, :, 0].double()).float()
    for k in range(1, 4):
        ze = (zf[:, k:k + 1].double() * e[None, :, k].double() + ze.double()).float()
    d_mine = (sq(zf)[:, None] + sq(e)[None, :]) - 2 * ze
    assert torch.equal(d_ref, d_mine), "distance rounding sequence differs from the documented one"
    print("  distance rounding sequence reproduced bit for bit on", d_ref.numel(), "distances")

    print("[G11] decode 4x64x64 -> 3x256x256")
    ref = ld.decode_first_stage(z)
    mine, _ = O.decode_first_stage(vsd, W.VQ_F4_256, z)
    check("decode_first_stage 256^2", ref, mine, 1e-4, 2e-4)
    g["decoded256"] = ref
    g["decoded256_noquant"] = ld.decode_first_stage(z, force_not_quantize=True)

    print("[G11] decode 3x32x32 -> 3x128x128 in fp32 (g6 holds this frame in fp16 only)")
    from ldm.models.autoencoder import VQModelInterface
    fs128 = VQModelInterface(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                             lossconfig=dict(target="torch.nn.Identity"))
    vsd128 = load_recipe(fs128, seed=0, prefix_check=W.vqmodel_param_shapes(W.VQ_F4))
    z128 = rnd(61, 1, 3, 32, 32)
    ref = fs128.decode(z128)
    mine, _ = O.decode_first_stage(vsd128, W.VQ_F4, z128)
    check("decode_first_stage 128^2", ref, mine, 1e-4, 2e-4)
    g["decoded128"] = ref
    save("g11_northstar.npz", **g)


# --------------------------------------------------------------------------- BASELINE configs[0] at the metric's shape
def gen_config0():
    """G13: BASELINE configs[0] -- DDIM 50 steps, batch 1 -- at the 64x64x4 latent, from the real reference's
    DDIMSampler.sample on the FR UNet's null-class token (DESIGN F8: a spatial-transformer UNet cannot run context=None),
    weights gain 0.25 as in the 32x32x3 test.  ~50 CPU UNet evaluations at 64x64: minutes, so it is generated here and not
    recomputed by the oracle on the GPU box."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    torch.set_grad_enabled(False)
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.NS_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=4, n_embed=16384, ddconfig=dict(W.VQ_F4_256["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True,
                         conditioning_key="crossattn", unet_config=unet_cfg, image_size=64, channels=4,
                         first_stage_key="image", log_every_t=200, monitor="val_loss_ema", **W.SCHEDULE)
    sched = O.register_schedule(**W.SCHEDULE)

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)

    g = {}
    print("[G13] configs[0]: DDIM-50, batch 1, 64x64x4, null-class token")
    usd = load_recipe(ld.model.diffusion_model, seed=0, gain=0.25, prefix_check=W.unet_param_shapes(W.NS_UNET))
    csd = load_recipe(ld.cond_stage_model, seed=0)
    uc = ld.cond_stage_model.uncond_embedding(torch.zeros(1, 1, dtype=torch.long))
    assert torch.equal(uc[0], csd["uncond_embedding.weight"])
    xT = rnd(0, 1, 4, 64, 64)
    ref, inter = CPUDDIM(ld).sample(S=50, batch_size=1, shape=[4, 64, 64], conditioning=uc, eta=0.0, x_T=xT, verbose=False,
                                    log_every_t=10)
    mine = O.ddim_sample(usd, W.NS_UNET, sched, 50, xT, cond=csd["uncond_embedding.weight"][None])
    check("DDIMSampler.sample S=50 @64x64x4, B=1", ref, mine, 1e-3, 1e-3)
    g["ns_ddim50"] = ref
    g["ns_ddim50_x_inter"] = torch.stack(inter["x_inter"])
    print("  |x| final", float(ref.abs().max()), "x_inter", len(inter["x_inter"]))

    print("[G13] configs[0] as worded: the UNCONDITIONAL LDM (cond_stage_config __is_unconditional__, AttentionBlock UNet)")
    ucfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.UNCOND_UNET))
    ldu = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config="__is_unconditional__", num_timesteps_cond=1,
                          unet_config=ucfg, image_size=64, channels=4, first_stage_key="image", log_every_t=200,
                          monitor="val_loss_ema", **W.SCHEDULE)
    assert ldu.model.conditioning_key is None
    usd_u = load_recipe(ldu.model.diffusion_model, seed=0, gain=0.25, prefix_check=W.unet_param_shapes(W.UNCOND_UNET))
    xe, te = rnd(130, 2, 4, 64, 64), torch.tensor([7, 640])
    ref_eps = ldu.apply_model(xe, te, None)
    check("unconditional UNet eps (B=2, 64x64x4)", ref_eps, O.unet_forward(usd_u, W.UNCOND_UNET, xe, te, None), 1e-4, 1e-4)
    g["uncond_eps"] = ref_eps
    # one AttentionBlock on its own (160 channels, 5 heads of 32, 8x8), the unit under the UNet
    from ldm.modules.diffusionmodules.openaimodel import AttentionBlock
    ab = AttentionBlock(160, num_heads=-1, num_head_channels=32)
    absd = load_recipe(ab, seed=11)
    xa = rnd(131, 2, 160, 8, 8)
    ref_ab = ab._forward(xa)
    check("AttentionBlock(160, heads 5)", ref_ab, O.attention_block(absd, "", xa, 5), 1e-5, 2e-5)
    g["attention_block"] = ref_ab
    ref_u, _ = CPUDDIM(ldu).sample(S=50, batch_size=1, shape=[4, 64, 64], conditioning=None, eta=0.0, x_T=xT, verbose=False)
    mine_u = O.ddim_sample(usd_u, W.UNCOND_UNET, sched, 50, xT, cond=None)
    check("unconditional DDIMSampler.sample S=50 @64x64x4, B=1", ref_u, mine_u, 1e-3, 1e-3)
    g["uncond_ddim50"] = ref_u
    print("  |x| final (unconditional)", float(ref_u.abs().max()))
    save("g13_config0.npz", **g)


# --------------------------------------------------------------------------- sampler options (ddim.py:112-203, ddpm.py:1049-1216)
class ShiftCorrector:
    """A deterministic stand-in for the (unshipped) score-corrector plugin: modify_score(model, e_t, x, t, c)."""

    def modify_score(self, model, e_t, x, t, c, strength=0.1):
        return e_t - strength * x * (t.float().view(-1, 1, 1, 1) / 1000.0)


def gen_options():
    """G12: the reference's own DDIMSampler.sample / LatentDiffusion.p_sample_loop with the options no shipped script sets
    (mask + x0 inpainting blend, temperature, quantize_x0 / quantize_denoised, score_corrector, clip_denoised), on seeded
    CPU noise.  The reference draws, per DDIM step, q_sample's randn_like (only with a mask) and then noise_like's randn
    (always): the same draws are replayed here in the same order and stored, so the GPU path can inject them."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    torch.set_grad_enabled(False)
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.FR_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True, conditioning_key="crossattn",
                         unet_config=unet_cfg, image_size=32, channels=3, first_stage_key="image", log_every_t=200,
                         monitor="val_loss_ema", **W.SCHEDULE)
    sched = O.register_schedule(**W.SCHEDULE)
    usd = load_recipe(ld.model.diffusion_model, seed=0, gain=0.25)
    load_recipe(ld.cond_stage_model, seed=0)
    vsd = load_recipe(ld.first_stage_model, seed=0)
    code = vsd["quantize.embedding.weight"]

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)

    sampler = CPUDDIM(ld)
    labels = torch.tensor([1, 6])
    c = ld.cond_stage_model.embedding(labels[:, None])
    uc = ld.cond_stage_model.uncond_embedding(torch.zeros(2, 1, dtype=torch.long))
    xT, x0 = rnd(51, 2, 3, 32, 32), 0.5 * rnd(52, 2, 3, 32, 32)
    mask = (rnd(53, 2, 1, 32, 32) > 0).float()
    S = 4
    g = {}

    def draws(seed, with_mask):
        torch.manual_seed(seed)
        mz, nz = [], []
        for _ in range(S):
            if with_mask:
                mz.append(torch.randn(xT.shape))
            nz.append(torch.randn(xT.shape))
        return (torch.stack(mz) if with_mask else None), torch.stack(nz)

    # (a) mask + x0, eta 1, temperature 0.7, guidance 3
    mz, nz = draws(7, True)
    torch.manual_seed(7)
    ref, _ = sampler.sample(S=S, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=1.0, x_T=xT, verbose=False, mask=mask,
                            x0=x0, temperature=0.7, unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
    mine = O.ddim_sample(usd, W.FR_UNET, sched, S, xT, cond=c, eta=1.0, scale=3.0, uncond=uc, noise=nz, mask=mask, x0=x0,
                         mask_noise=mz, temperature=0.7)
    check("ddim mask/x0 + temperature + cfg", ref, mine, 1e-4, 1e-4)
    g["ddim_mask_temp_cfg"], g["mask_noise"], g["step_noise"] = ref, mz, nz
    # (b) quantize_x0, eta 0
    torch.manual_seed(8)
    ref, inter = sampler.sample(S=S, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False,
                                quantize_x0=True, log_every_t=1)
    mine = O.ddim_sample(usd, W.FR_UNET, sched, S, xT, cond=c, quantize_codebook=code)
    check("ddim quantize_x0", ref, mine, 1e-4, 1e-4)
    g["ddim_quantize"], g["ddim_quantize_pred_x0"] = ref, inter["pred_x0"][-1]
    # (c) score corrector, eta 0, guidance 3
    corr = ShiftCorrector()
    torch.manual_seed(9)
    ref, _ = sampler.sample(S=S, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False,
                            score_corrector=corr, corrector_kwargs=dict(strength=0.2), unconditional_guidance_scale=3.0,
                            unconditional_conditioning=uc)
    mine = O.ddim_sample(usd, W.FR_UNET, sched, S, xT, cond=c, scale=3.0, uncond=uc,
                         score_fn=lambda e, x, t: corr.modify_score(None, e, x, t, None, strength=0.2))
    check("ddim score_corrector + cfg", ref, mine, 1e-4, 1e-4)
    g["ddim_corrector_cfg"] = ref
    # (d) ancestral loop: clip_denoised + quantize_denoised + mask, 3 steps.  p_sample draws noise_like first, the mask
    # blend's q_sample second (ddpm.py:1199-1208)
    T_ = 3
    torch.manual_seed(11)
    pn, pm = [], []
    for _ in range(T_):
        pn.append(torch.randn(xT.shape))
        pm.append(torch.randn(xT.shape))
    ld.clip_denoised = True
    torch.manual_seed(11)
    ref = ld.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=T_, verbose=False, quantize_denoised=True, mask=mask, x0=x0)
    ld.clip_denoised = False
    mine = O.p_sample_loop(usd, W.FR_UNET, sched, xT, cond=c, timesteps=T_, noise=pn, clip_denoised=True,
                           quantize_codebook=code, mask=mask, x0=x0, mask_noise=pm)
    check("p_sample_loop clip + quantize + mask", ref, mine, 1e-4, 1e-4)
    g["ddpm_clip_quant_mask"], g["ddpm_noise"], g["ddpm_mask_noise"] = ref, torch.stack(pn), torch.stack(pm)
    save("g12_sampler_options.npz", **g)


def gen_variants():
    """G14: reference kwargs the shipped YAMLs do not set but checkpoints in the wild do -- attention heads that are not 32 wide
    (num_heads = 4 -> 40 / 80, num_head_channels = 64), from the real UNetModel -- and `use_original_steps` of the sampler from
    the real talking_face DDIMSampler.p_sample_ddim (ddim2cond.py:158-195; the face_reenactment copy raises AttributeError there)."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    torch.set_grad_enabled(False)
    g = {}
    for tag, cfg in (("h40", W.H40_UNET), ("h64", W.H64_UNET)):
        m = UNetModel(**cfg)
        sd = load_recipe(m, seed=0, prefix_check=W.unet_param_shapes(cfg))
        x, t, ctx = rnd(150, 2, 3, 16, 16), torch.tensor([11, 870]), rnd(151, 2, 1, 512)
        ref = m(x, t, context=ctx)
        check(f"UNet eps, {tag} ({cfg.get('num_heads', cfg.get('num_head_channels'))})", ref, O.unet_forward(sd, cfg, x, t, ctx), 2e-5, 2e-5)
        g[tag + "_eps"] = ref
        ctx3 = rnd(152, 2, 3, 512)                 # a context of several tokens: the cross attention is live
        ref3 = m(x, t, context=ctx3)
        check(f"UNet eps, {tag}, 3 context tokens", ref3, O.unet_forward(sd, cfg, x, t, ctx3), 2e-5, 2e-5)
        g[tag + "_eps_L3"] = ref3
    # ---- the class-conditional ('adm') UNet with use_scale_shift_norm, num_classes, use_new_attention_order (QKVAttention)
    m = UNetModel(**W.ADM_UNET)
    sd = load_recipe(m, seed=0, prefix_check=W.unet_param_shapes(W.ADM_UNET))
    x, t, y = rnd(153, 2, 3, 16, 16), torch.tensor([3, 512]), torch.tensor([7, 2])
    ref = m(x, t, y=y)
    check("UNet eps, adm (scale-shift norm, 10 classes, QKVAttention)", ref, O.unet_forward(sd, W.ADM_UNET, x, t, None, y=y), 2e-5, 2e-5)
    g["adm_eps"] = ref
    # ---- use_original_steps: the real sampler class of the talking-face tree around a stub model that returns a fixed eps
    for k in [k for k in sys.modules if k == "ldm" or k.startswith("ldm.") or k.startswith("taming")]:
        del sys.modules[k]
    ref_shims.install("talking_face")
    from ldm.models.diffusion.ddim2cond import DDIMSampler
    sched = O.register_schedule(**W.SCHEDULE)
    x, eps = rnd(160, 2, 3, 32, 32), rnd(161, 2, 3, 32, 32)

    class Stub:
        num_timesteps = 1000
        device = torch.device("cpu")
        betas, alphas_cumprod, alphas_cumprod_prev = sched["betas"], sched["alphas_cumprod"], sched["alphas_cumprod_prev"]
        sqrt_one_minus_alphas_cumprod = sched["sqrt_one_minus_alphas_cumprod"]
        parameterization = "eps"

        def apply_model(self, x_, t_, c12, c34):
            return eps

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)
    for eta in (0.0, 1.0):
        smp = CPUDDIM(Stub())
        smp.make_schedule(50, ddim_eta=eta, verbose=False)
        tabs = O.ddim_original_tables(sched, eta)
        for index in (0, 1, 437, 999):
            torch.manual_seed(1000 + index)
            xp, p0 = smp.p_sample_ddim(x, {"class_label_&_audio": None, "motion_&_id": None}, torch.full((2,), index), index,
                                       use_original_steps=True)
            torch.manual_seed(1000 + index)
            nz = torch.randn(x.shape)
            mine = O.p_sample_ddim_original(x, eps, index, tabs, nz)
            check(f"p_sample_ddim(use_original_steps) eta={eta:g} index={index}", xp, mine[0], 1e-6, 1e-6)
            check(f"   pred_x0", p0, mine[1], 1e-6, 1e-6)
            g[f"orig_eta{eta:g}_i{index}_x_prev"], g[f"orig_eta{eta:g}_i{index}_pred_x0"] = xp, p0
            g[f"orig_eta{eta:g}_i{index}_noise"] = nz
    save("g14_variants.npz", **g)


def gen_updown():
    """G15: `resblock_updown=True` from the real UNetModel (openaimodel.py:570-584,660-674): the shipped spatial-transformer UNet (one ResBlock per
    level) with ResBlock(down=True) / ResBlock(up=True) between its three levels, and the class-conditional UNet with use_scale_shift_norm on top -- the latter
    also inside a real LatentDiffusion(conditioning_key='adm') under the real DDIMSampler.sample (plain, CFG) and p_sample_loop."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    torch.set_grad_enabled(False)
    g = {}
    m = UNetModel(**W.UPDOWN_UNET)
    sd = load_recipe(m, seed=0, prefix_check=W.unet_param_shapes(W.UPDOWN_UNET))
    x, t, ctx = rnd(170, 2, 3, 32, 32), torch.tensor([5, 640]), rnd(171, 2, 1, 512)
    ref = m(x, t, context=ctx)
    check("UNet eps, resblock_updown (spatial transformers, 32x32)", ref, O.unet_forward(sd, W.UPDOWN_UNET, x, t, ctx), 2e-5, 2e-5)
    g["ud_eps"] = ref
    m = UNetModel(**W.UPDOWN_ADM_UNET)
    sd = load_recipe(m, seed=0, prefix_check=W.unet_param_shapes(W.UPDOWN_ADM_UNET))
    x, t, y = rnd(173, 2, 3, 16, 16), torch.tensor([3, 512]), torch.tensor([7, 2])
    ref = m(x, t, y=y)
    check("UNet eps, resblock_updown + scale-shift norm (adm)", ref, O.unet_forward(sd, W.UPDOWN_ADM_UNET, x, t, None, y=y), 2e-5, 2e-5)
    g["ud_adm_eps"] = ref
    # ---- the samplers around a class-conditional model: a real LatentDiffusion with conditioning_key 'adm' (the class labels go
    # through apply_model as c_crossattn = [y] and reach the UNet as y, ddpm.py:893-994,1417-1419), the real DDIMSampler.sample
    # (plain and with classifier-free guidance, ddim.py:164-177: y doubled as [uncond | cond]) and p_sample_loop
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.UPDOWN_ADM_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True,
                         conditioning_key="adm", unet_config=unet_cfg, image_size=16, channels=3,
                         first_stage_key="image", log_every_t=200, monitor="val_loss_ema", **W.SCHEDULE)
    assert ld.model.conditioning_key == "adm"
    usd = load_recipe(ld.model.diffusion_model, seed=0, gain=0.25, prefix_check=W.unet_param_shapes(W.UPDOWN_ADM_UNET))   # (gain as in g5)
    sched = O.register_schedule(**W.SCHEDULE)

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)
    xT, uy = rnd(176, 2, 3, 16, 16), torch.tensor([0, 0])
    ref, _ = CPUDDIM(ld).sample(S=4, batch_size=2, shape=[3, 16, 16], conditioning=y, eta=0.0, x_T=xT, verbose=False)
    check("adm DDIMSampler.sample S=4", ref, O.ddim_sample(usd, W.UPDOWN_ADM_UNET, sched, 4, xT, cond=y), 1e-4, 1e-4)
    g["adm_ddim4"] = ref
    ref, _ = CPUDDIM(ld).sample(S=4, batch_size=2, shape=[3, 16, 16], conditioning=y, eta=0.0, x_T=xT, verbose=False,
                                unconditional_guidance_scale=3.0, unconditional_conditioning=uy)
    check("adm DDIMSampler.sample S=4, CFG 3", ref, O.ddim_sample(usd, W.UPDOWN_ADM_UNET, sched, 4, xT, cond=y, scale=3.0, uncond=uy),
          1e-4, 1e-4)
    g["adm_ddim4_cfg3"] = ref
    torch.manual_seed(6)
    ref = ld.p_sample_loop(y, (2, 3, 16, 16), x_T=xT, timesteps=3, verbose=False)
    torch.manual_seed(6)
    nz = [torch.randn(xT.shape) for _ in range(3)]
    check("adm p_sample_loop T=3", ref, O.p_sample_loop(usd, W.UPDOWN_ADM_UNET, sched, xT, cond=y, timesteps=3, noise=nz), 1e-4, 1e-4)
    g["adm_ddpm3"], g["adm_ddpm3_noise"] = ref, torch.stack(nz)
    save("g15_updown.npz", **g)


def gen_sdedit():
    """G16: the sampler's img2img pair from the real talking-face DDIMSampler (ddim2cond.py:198-250) around the real LatentDiffusion:
    stochastic_encode (q_sample on the DDIM subsequence / on the model's own schedule) and decode (the DDIM updates of the first
    t_start entries of the subsequence), eta 0 and eta 1 with the reference's own seeded draws replayed and stored."""
    from tools import ref_shims
    ref_shims.install("talking_face")
    from ldm.models.diffusion.ddim2cond import DDIMSampler
    from ldm.models.diffusion.ddpm2cond import LatentDiffusion
    torch.set_grad_enabled(False)
    sched = O.register_schedule(**W.SCHEDULE)
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.TF_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    c1_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder",
                  params=dict(embed_dim=256, n_classes=8, key="class_label", p_uncond=0.2))
    c2_cfg = dict(target="ldm.modules.encoders.modules.Conv1DTemporalAttention",
                  params=dict(seq_len=3, subspace_dim=768, subspace2hidden=False))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config_1=c1_cfg, cond_stage_config_2=c2_cfg,
                         num_timesteps_cond=1, cond_stage_key_1="class_label", cond_stage_key_2="audio",
                         cond_stage_trainable=True, conditioning_key="crossattn", unet_config=unet_cfg,
                         image_size=32, channels=3, first_stage_key="image", log_every_t=200,
                         monitor="val_loss_ema", **W.SCHEDULE)

    class CPUDDIM(DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)
    usd = load_recipe(ld.model.diffusion_model, seed=0, gain=0.25, prefix_check=W.unet_param_shapes(W.TF_UNET))
    g = {}
    S = 5
    x0, nz = rnd(180, 2, 3, 32, 32), rnd(181, 2, 3, 32, 32)
    c12, c34 = rnd(182, 2, 1, 1024), rnd(183, 2, 6, 32, 32)
    cond = {"class_label_&_audio": c12, "motion_&_id": c34}
    smp = CPUDDIM(ld)
    smp.make_schedule(S, ddim_eta=0.0, verbose=False)
    t = torch.tensor([3, 1])
    enc = smp.stochastic_encode(x0, t, noise=nz)
    check("stochastic_encode (DDIM subsequence)", enc, O.stochastic_encode(sched, S, x0, t, nz), 1e-6, 1e-6)
    g["enc"] = enc
    t_o = torch.tensor([640, 7])
    enc_o = smp.stochastic_encode(x0, t_o, use_original_steps=True, noise=nz)
    check("stochastic_encode (original steps)", enc_o, O.stochastic_encode(sched, S, x0, t_o, nz, use_original_steps=True), 1e-6, 1e-6)
    g["enc_orig"] = enc_o
    t3 = torch.tensor([2, 2])                    # encode to index 2, decode the first 3 entries: an SDEdit edit of strength 3 / 5
    x_lat = smp.stochastic_encode(x0, t3, noise=nz)
    dec = smp.decode(x_lat, cond, 3)
    mine = O.ddim_decode(usd, W.TF_UNET, sched, S, O.stochastic_encode(sched, S, x0, t3, nz), 3, cond=c12, c_concat=c34)
    check("decode t_start=3 of S=5, eta 0", dec, mine, 1e-4, 1e-4)
    g["dec3"] = dec
    smp1 = CPUDDIM(ld)
    smp1.make_schedule(S, ddim_eta=1.0, verbose=False)
    torch.manual_seed(8)
    dec1 = smp1.decode(x_lat, cond, 3)
    torch.manual_seed(8)
    dn = [torch.randn(x0.shape) for _ in range(3)]
    mine = O.ddim_decode(usd, W.TF_UNET, sched, S, x_lat, 3, cond=c12, c_concat=c34, eta=1.0, noise=dn)
    check("decode t_start=3 of S=5, eta 1", dec1, mine, 1e-4, 1e-4)
    g["dec3_eta1"], g["dec3_eta1_noise"] = dec1, torch.stack(dn)
    save("g16_sdedit.npz", **g)


def gen_manip():
    """G17: DDIMSampler.latent_manipulation of the driver script face_reenactment/latent_manipulation.py (:420-490; importable under
    the same stand-ins as compute_latents.py): DDIM inversion of x0 under the SOURCE label's conditioning, regeneration under the
    TARGET label's -- the emotion edit -- on the real LatentDiffusion, plain and with classifier-free guidance."""
    from tools import ref_shims
    ref_shims.install("face_reenactment")
    from ldm.models.diffusion.ddpm import LatentDiffusion
    import latent_manipulation as lm
    torch.set_grad_enabled(False)
    unet_cfg = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.FR_UNET))
    fs_cfg = dict(target="ldm.models.autoencoder.VQModelInterface",
                  params=dict(embed_dim=3, n_embed=16384, ddconfig=dict(W.VQ_F4["ddconfig"]),
                              lossconfig=dict(target="torch.nn.Identity")))
    cond_cfg = dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                    params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2))
    ld = LatentDiffusion(first_stage_config=fs_cfg, cond_stage_config=cond_cfg, num_timesteps_cond=1,
                         cond_stage_key="class_label", cond_stage_trainable=True,
                         conditioning_key="crossattn", unet_config=unet_cfg, image_size=32, channels=3,
                         first_stage_key="image", log_every_t=200, monitor="val_loss_ema", **W.SCHEDULE)
    sched = O.register_schedule(**W.SCHEDULE)
    usd = load_recipe(ld.model.diffusion_model, seed=0, gain=0.25, prefix_check=W.unet_param_shapes(W.FR_UNET))
    load_recipe(ld.cond_stage_model, seed=0)
    c_src = ld.cond_stage_model.embedding(torch.tensor([[1], [6]]))
    c_trg = ld.cond_stage_model.embedding(torch.tensor([[3], [0]]))
    uc = ld.cond_stage_model.uncond_embedding(torch.zeros(2, 1, dtype=torch.long))

    class CPUManip(lm.DDIMSampler):
        def register_buffer(self, n, a):
            setattr(self, n, a)
    smp = CPUManip(ld)
    x0 = rnd(190, 2, 3, 32, 32)
    g = {}
    for tag, scale in (("cfg1", 1.0), ("cfg3", 3.0)):
        img, xlat, _ = smp.latent_manipulation(c_src, c_trg, S=4, batch_size=2, shape=[3, 32, 32], x0=x0, eta=0.0, verbose=False,
                                               strength=0.5, unconditional_guidance_scale=scale,
                                               unconditional_conditioning=uc if scale != 1.0 else None)
        mimg, mlat = O.ddim_invert_and_regenerate(usd, W.FR_UNET, sched, 4, x0, c_src, strength=0.5, scale=scale,
                                                  uncond=uc if scale != 1.0 else None, cond_trg=c_trg)
        check(f"manipulation: inverted latent {tag}", xlat, mlat, 1e-4, 1e-4)
        check(f"manipulation: edited {tag}", img, mimg, 1e-4, 1e-4)
        g[f"xlat_{tag}"], g[f"img_{tag}"] = xlat, img
    same, _, _ = smp.latent_manipulation(c_src, c_src, S=4, batch_size=2, shape=[3, 32, 32], x0=x0, eta=0.0, verbose=False, strength=0.5)
    print("  edit moves the image by", float((g["img_cfg1"] - same).abs().max()))
    g["c_src"], g["c_trg"], g["uc"] = c_src, c_trg, uc
    save("g17_manipulation.npz", **g)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", choices=["face_reenactment", "talking_face", "train", "diffclip", "northstar", "options", "config0", "variants", "updown", "sdedit", "manip"])
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    if a.tree == "train":
        gen_train()
    elif a.tree == "diffclip":
        gen_diffclip()
    elif a.tree == "face_reenactment":
        gen_fr()
    elif a.tree == "talking_face":
        gen_tf()
    elif a.tree == "northstar":
        gen_northstar()
    elif a.tree == "options":
        gen_options()
    elif a.tree == "config0":
        gen_config0()
    elif a.tree == "variants":
        gen_variants()
    elif a.tree == "updown":
        gen_updown()
    elif a.tree == "sdedit":
        gen_sdedit()
    elif a.tree == "manip":
        gen_manip()
    else:
        for tree in ("face_reenactment", "talking_face", "train", "diffclip", "northstar", "options", "config0", "variants", "updown", "sdedit", "manip"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--tree", tree], cwd=ROOT)
