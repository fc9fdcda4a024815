"""GPU: samplers, first stage and the talking-face path against the reference's outputs (golden fixtures),
driven through the reference's own call surface (instantiate_from_config -> LatentDiffusion -> DDIMSampler)."""
import numpy as np
import pytest
import torch

from conftest import golden, rnd
from helpers import make_fr_model, make_tf_model
from oracle import ldm_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.asarray(a))


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.float().cpu(), torch.as_tensor(np.asarray(b)).float(), rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def fr():
    return make_fr_model(gain=0.25)


def _cond(m, labels=(1, 6)):
    lab = torch.tensor(labels, device="cuda")[:, None]
    return m.cond_stage_model.embedding(lab), m.cond_stage_model.uncond_embedding(torch.zeros_like(lab))


def test_schedule_buffers_match_reference(fr):
    g = golden("g1_schedules.npz")
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "posterior_mean_coef1",
              "posterior_mean_coef2", "posterior_log_variance_clipped"):
        assert np.array_equal(getattr(fr, k).cpu().numpy(), g[k]), k


def test_ddim_sample_S4_and_graph_replay(fr):
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g5_sampling_fr.npz")
    c, _ = _cond(fr)
    xT = rnd(51, 2, 3, 32, 32).cuda()
    s = DDIMSampler(fr)
    out, inter = s.sample(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False)
    # 4 chained UNet evaluations with |x| growing to ~48: relative tolerance
    close(out, g["sample_S4"], 1.5e-4, 1.5e-4)
    assert len(inter["x_inter"]) >= 2 and inter["pred_x0"][-1].shape == out.shape
    out_g, _ = s.sample(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False,
                        use_graph=True)
    assert torch.equal(out, out_g), "hipGraph replay must be bitwise identical to eager launches"
    out_g2, _ = s.sample(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False,
                         use_graph=True)
    assert torch.equal(out, out_g2), "a cached graph replays from a clean state"


def test_f16x2_range_fallback_inside_a_graphed_sampling_run():
    """The F16X2 range flags in a SAMPLING run under hipGraph replay (no host read inside the loop): a checkpoint whose attn1.to_k
    weights of one block are 3000 x larger (to_q as much smaller: same logits) finishes its DDIM run, the sampler reads the flags
    once, warns, re-plans THAT site in bf16x3 and repeats the run -- the same run: with eta = 1 and no x_T / noise given, start
    noise and per-step noise come out of the restored generator again, so the result is bit for bit what a model that was told
    about the site up front (deny_f16x2) samples from the same seed, graph or eager start noise alike."""
    import warnings
    from dsml_thesis_amd.ddim import DDIMSampler
    site = "input_blocks.1.1.transformer_blocks.0.attn1"

    def model():
        m = make_fr_model(gain=0.25)
        sd = m.model.diffusion_model.state_dict()
        key = site + ".to_k.weight"
        sd[key] = sd[key] * 3000.0
        sd[key.replace("to_k", "to_q")] = sd[key.replace("to_k", "to_q")] / 3000.0
        m.model.diffusion_model.load_state_dict(sd, strict=True)
        return m
    xT = rnd(52, 16, 3, 32, 32).cuda()
    m = model()
    c, _ = _cond(m, labels=tuple(i % 7 for i in range(16)))       # (class ids of the 8-class embedder)
    unet = m.model.diffusion_model
    assert unet.f16x2
    fired = []
    with pytest.warns(RuntimeWarning, match="F16X2"):
        out, _ = DDIMSampler(m).sample(S=4, batch_size=16, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False, use_graph=True,
                                       callback=fired.append)
    st = unet.arithmetic_status()
    assert st["f16x2"] and st["denied"] == [site] and torch.isfinite(out).all(), st
    assert fired == [0, 1, 2, 3] * 2                               # (documented: callbacks fire during the discarded pass too)
    m0 = model()
    m0.model.diffusion_model.deny_f16x2([site])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ref, _ = DDIMSampler(m0).sample(S=4, batch_size=16, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False, use_graph=False)
    assert torch.equal(out, ref)
    # eta = 1, nothing pinned: the repeated pass must replay the first pass's draws
    m1 = model()
    torch.manual_seed(1234)
    with pytest.warns(RuntimeWarning, match="F16X2"):
        out1, _ = DDIMSampler(m1).sample(S=4, batch_size=16, shape=[3, 32, 32], conditioning=c, eta=1.0, verbose=False, use_graph=True)
    torch.manual_seed(1234)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ref1, _ = DDIMSampler(m0).sample(S=4, batch_size=16, shape=[3, 32, 32], conditioning=c, eta=1.0, verbose=False, use_graph=True)
    assert torch.isfinite(out1).all() and torch.equal(out1, ref1)


@pytest.mark.parametrize("latent", [32, 64])
def test_config0_ddim50_batch1_end_to_end(fr, latent):
    """BASELINE configs[0]: DDIM 50 steps, batch 1, the 'unconditional' plumbing case.  A spatial-transformer UNet fed
    context=None raises in the reference itself (DESIGN F8), so the unconditional run is the FR UNet on its null-class
    token (`uncond_embedding`, what sample_affectnet.py:93-94 feeds as the unconditional branch).  End to end at S = 50,
    B = 1: finite, hipGraph replay == eager launches bit for bit, and (32x32x3) equal to the oracle's 50-step chain."""
    from dsml_thesis_amd import synth
    from dsml_thesis_amd.ddim import DDIMSampler
    m = fr if latent == 32 else make_fr_model(gain=0.25, unet=synth.NS_UNET, vq=synth.VQ_F4_256)
    ch = 3 if latent == 32 else 4
    lab = torch.zeros(1, 1, dtype=torch.long, device="cuda")
    uc = m.cond_stage_model.uncond_embedding(lab)                                   # (1,1,512)
    xT = rnd(0, 1, ch, latent, latent).cuda()
    s = DDIMSampler(m)
    out, inter = s.sample(S=50, batch_size=1, shape=[ch, latent, latent], conditioning=uc, eta=0.0, x_T=xT, verbose=False)
    assert s.ddim_timesteps.shape == (50,) and int(s.ddim_timesteps[0]) == 1 and int(s.ddim_timesteps[-1]) == 981
    assert torch.isfinite(out).all() and out.shape == (1, ch, latent, latent)
    out_g, _ = s.sample(S=50, batch_size=1, shape=[ch, latent, latent], conditioning=uc, eta=0.0, x_T=xT, verbose=False,
                        use_graph=True)
    assert torch.equal(out, out_g), "hipGraph replay must be bitwise identical to eager launches"
    if latent == 64:
        # the reference's own 50-step chain at the metric's shape (tests/golden/g13_config0.npz, tools/make_golden.py --tree
        # config0: DDIMSampler.sample of the real reference, ~50 CPU evaluations at 64x64, generated in the build container)
        g = golden("g13_config0.npz")
        ref = T(g["ns_ddim50"])
        d, top = (out.cpu() - ref).abs().max().item(), ref.abs().max().item()
        print(f"config0 64x64x4: max |diff| vs the reference after 50 steps {d:.3e} (|x| up to {top:.2f})")
        # 50 chained evaluations with random weights: |x| grows to ~350 and every element carries the trajectory's error, so
        # the bound is relative to the tensor's scale: 1e-5 of max |x| (measured 8e-7; the oracle restatement itself 7e-7)
        assert d <= 1e-5 * top, (d, top)
    if latent == 32:
        usd = W.synth_state_dict(W.unet_param_shapes(W.FR_UNET), gain=0.25)
        ucw = torch.from_numpy(W.synth_tensor("uncond_embedding.weight", (1, 512)))
        ref = O.ddim_sample(usd, W.FR_UNET, O.register_schedule(**W.SCHEDULE), 50, xT.cpu(), cond=ucw[None])
        d = (out.cpu() - ref).abs().max().item()
        print(f"config0 32x32x3: max |diff| vs the oracle after 50 steps {d:.3e} (|x| up to {ref.abs().max().item():.2f})")
        close(out, ref, 1e-3, 1e-3)
        img = m.decode_first_stage(out)
        assert img.shape == (1, 3, 128, 128) and torch.isfinite(img).all()


def test_config0_unconditional_ldm_ddim50_batch1_against_the_reference():
    """BASELINE configs[0] exactly as worded: an UNCONDITIONAL 64x64x4-latent LDM (cond_stage_config "__is_unconditional__" ->
    conditioning_key None -> diffusion_model(x, t), ddpm.py:443-444,1405-1406), DDIM 50 steps, batch 1, through the reference's
    own surface (instantiate -> LatentDiffusion -> DDIMSampler.sample(conditioning=None)) -- against the real reference's
    50-step chain (g13 `uncond_ddim50`); hipGraph replay == eager launches; decode to a 256x256 frame."""
    from helpers import make_uncond_model
    from dsml_thesis_amd.ddim import DDIMSampler
    from dsml_thesis_amd import lib as L
    g = golden("g13_config0.npz")
    m = make_uncond_model(gain=0.25)
    assert m.model.conditioning_key is None and m.cond_stage_model is None
    xT = rnd(0, 1, 4, 64, 64).cuda()
    s = DDIMSampler(m)
    out, _ = s.sample(S=50, batch_size=1, shape=[4, 64, 64], conditioning=None, eta=0.0, x_T=xT, verbose=False)
    ref = T(g["uncond_ddim50"])
    d, top = (out.cpu() - ref).abs().max().item(), ref.abs().max().item()
    print(f"config0 unconditional 64x64x4: max |diff| vs the reference after 50 steps {d:.3e} (|x| up to {top:.2f})")
    assert d <= 1e-5 * top, (d, top)                       # (measured 9.4e-7 of max |x|)
    out_g, _ = s.sample(S=50, batch_size=1, shape=[4, 64, 64], conditioning=None, eta=0.0, x_T=xT, verbose=False, use_graph=True)
    assert torch.equal(out, out_g)
    img = m.decode_first_stage(out)
    assert img.shape == (1, 3, 256, 256) and torch.isfinite(img).all()
    # apply_model / p_sample_loop take cond = None too (ddpm.py:1405-1406)
    e = m.apply_model(xT, torch.tensor([981], device="cuda"), None)
    assert e.shape == xT.shape and torch.isfinite(e).all()
    z = m.p_sample_loop(None, (1, 4, 64, 64), x_T=xT, timesteps=3, verbose=False)
    assert torch.isfinite(z).all()
    # a conditional model still refuses conditioning=None
    with pytest.raises(L.LdmkError, match="conditioning is required"):
        DDIMSampler(make_fr_model(gain=0.25)).sample(S=4, batch_size=1, shape=[3, 32, 32], conditioning=None, verbose=False)


def test_concat_conditioning_key_equals_channel_concatenated_input(fr):
    """DiffusionWrapper conditioning_key='concat' (ddpm.py:1407-1409): `diffusion_model(cat([x] + c_concat, 1), t)`.  The UNet's
    first convolution reads both tensors (nothing is concatenated in memory): equal bit for bit to feeding the concatenated
    tensor, through apply_model and through DDIMSampler.sample."""
    from dsml_thesis_amd import synth
    from dsml_thesis_amd.ddpm import LatentDiffusion
    from dsml_thesis_amd.ddim import DDIMSampler
    cfg = synth.uncond_config(dict(synth.UNCOND_UNET, image_size=32, in_channels=7, out_channels=4), synth.VQ_F4_256)
    cfg.update(conditioning_key="concat", cond_stage_config="__is_first_stage__", channels=4, image_size=32)
    m = LatentDiffusion(**cfg)
    synth.load_recipe(m.model.diffusion_model, gain=0.25)
    m = m.cuda().eval()
    x, cc, t = rnd(140, 2, 4, 32, 32).cuda(), rnd(141, 2, 3, 32, 32).cuda(), torch.tensor([5, 700], device="cuda")
    a = m.apply_model(x, t, cc)
    b = m.model.diffusion_model(torch.cat([x, cc], 1), t)
    assert torch.equal(a, b)
    out, _ = DDIMSampler(m).sample(S=4, batch_size=2, shape=[4, 32, 32], conditioning=cc, eta=0.0, x_T=x, verbose=False)
    assert out.shape == (2, 4, 32, 32) and torch.isfinite(out).all()


def test_concat_conditioning_with_guidance_single_step_equals_the_loop():
    """`p_sample_ddim` with 'concat' conditioning AND classifier-free guidance: the halves of the doubled batch differ in the
    concat tensor, [uncond | cond] (the reference's p_sample_ddim hands any `c` to apply_model, ddim.py:170-177; round 4 built the
    branch into ddim_sampling only and p_sample_ddim raised a TypeError).  Chained single steps == the sampling loop, bit for bit."""
    from dsml_thesis_amd import synth
    from dsml_thesis_amd.ddpm import LatentDiffusion
    from dsml_thesis_amd.ddim import DDIMSampler
    cfg = synth.uncond_config(dict(synth.UNCOND_UNET, image_size=32, in_channels=7, out_channels=4), synth.VQ_F4_256)
    cfg.update(conditioning_key="concat", cond_stage_config="__is_first_stage__", channels=4, image_size=32)
    m = LatentDiffusion(**cfg)
    synth.load_recipe(m.model.diffusion_model, gain=0.25)
    m = m.cuda().eval()
    x, cc, ucc = rnd(142, 2, 4, 32, 32).cuda(), rnd(143, 2, 3, 32, 32).cuda(), rnd(144, 2, 3, 32, 32).cuda()
    s = DDIMSampler(m)
    out, _ = s.sample(S=4, batch_size=2, shape=[4, 32, 32], conditioning=cc, eta=0.0, x_T=x, verbose=False,
                      unconditional_guidance_scale=3.0, unconditional_conditioning=ucc)
    img = x
    for i, step in enumerate(np.flip(s.ddim_timesteps)):
        ts = torch.full((2,), int(step), device="cuda", dtype=torch.long)
        img, _ = s.p_sample_ddim(img, cc, ts, index=len(s.ddim_timesteps) - i - 1, unconditional_guidance_scale=3.0,
                                 unconditional_conditioning=ucc)
    assert torch.isfinite(out).all() and torch.equal(img, out)
    # the guidance really used the unconditional concat tensor
    out_same, _ = s.sample(S=4, batch_size=2, shape=[4, 32, 32], conditioning=cc, eta=0.0, x_T=x, verbose=False,
                           unconditional_guidance_scale=3.0, unconditional_conditioning=cc)
    assert not torch.equal(out, out_same)


def test_use_original_steps_walks_the_models_own_schedule(fr):
    """`use_original_steps` (ddim.py:127,133-134,183-186; ddim2cond.py:175-178): index counts the model's 1000 timesteps and the
    update takes alphas_cumprod / alphas_cumprod_prev / sqrt_one_minus_alphas_cumprod / eta sqrt((1 - a_prev) / (1 - a) (1 - a / a_prev)).
    One step against that formula in float64 on the UNet's own eps; the loop (`ddim_use_original_steps=True, timesteps=3`: steps 2,
    1, 0, ddim.py:133) against chained single steps bit for bit, eager and hipGraph."""
    from dsml_thesis_amd.ddim import DDIMSampler
    c, uc = _cond(fr)
    s = DDIMSampler(fr)
    s.make_schedule(50, ddim_eta=1.0, verbose=False)
    x = rnd(145, 2, 3, 32, 32).cuda()
    nz = rnd(146, 2, 3, 32, 32).cuda()
    idx = 437
    ts = torch.full((2,), idx, device="cuda", dtype=torch.long)
    xp, px0 = s.p_sample_ddim(x, c, ts, index=idx, use_original_steps=True, noise=nz)
    e = fr.apply_model(x, ts, c).double()
    a, ap = fr.alphas_cumprod[idx].double(), fr.alphas_cumprod_prev[idx].double()
    sig = torch.sqrt((1 - ap) / (1 - a) * (1 - a / ap))
    p0 = (x.double() - torch.sqrt(1 - a) * e) / torch.sqrt(a)
    ref = torch.sqrt(ap) * p0 + torch.sqrt(1 - ap - sig ** 2) * e + sig * nz.double()
    torch.testing.assert_close(px0.double(), p0, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(xp.double(), ref, rtol=2e-5, atol=2e-5)
    # ... and against the REAL talking-face sampler's outputs (g14: ddim2cond.DDIMSampler.p_sample_ddim(use_original_steps=True) around a
    # model that returns a fixed eps): the update kernel on the original-step tables, eta 0 and 1, four of the 1000 timesteps
    g = golden("g14_variants.npz")
    xs, ef = rnd(160, 2, 3, 32, 32).cuda(), rnd(161, 2, 3, 32, 32).cuda()
    real_apply = fr.apply_model
    try:
        fr.apply_model = lambda *a_, **k_: ef.clone()
        for eta in (0.0, 1.0):
            s.make_schedule(50, ddim_eta=eta, verbose=False)
            for i in (0, 1, 437, 999):
                nz_i = torch.from_numpy(g[f"orig_eta{eta:g}_i{i}_noise"]).cuda()
                xp_i, p0_i = s.p_sample_ddim(xs, c, torch.full((2,), i, device="cuda", dtype=torch.long), index=i, use_original_steps=True,
                                             noise=nz_i if eta else None)
                # (pred_x0 divides by sqrt(a_t) -- 0.0098 at index 999 -- so the bound scales with the tensor: 2e-6 of max |x|)
                top = float(np.abs(g[f"orig_eta{eta:g}_i{i}_pred_x0"]).max())
                close(p0_i, g[f"orig_eta{eta:g}_i{i}_pred_x0"], 2e-6, 2e-6 * max(1.0, top))
                close(xp_i, g[f"orig_eta{eta:g}_i{i}_x_prev"], 2e-6, 2e-6 * max(1.0, 0.02 * top))
    finally:
        del fr.apply_model
    assert fr.apply_model.__func__ is type(fr).apply_model
    s.make_schedule(50, ddim_eta=0.0, verbose=False)
    out, inter = s.ddim_sampling(c, (2, 3, 32, 32), x_T=x, ddim_use_original_steps=True, timesteps=3)
    img = x
    for i in (2, 1, 0):
        img, _ = s.p_sample_ddim(img, c, torch.full((2,), i, device="cuda", dtype=torch.long), index=i, use_original_steps=True)
    assert torch.equal(out, img)
    out_g, _ = s.ddim_sampling(c, (2, 3, 32, 32), x_T=x, ddim_use_original_steps=True, timesteps=3, use_graph=True)
    assert torch.equal(out, out_g)
    # `timesteps` on the DDIM subsequence: the first int(min(t / S, 1) S) - 1 entries (ddim.py:129-131)
    sub, _ = s.ddim_sampling(c, (2, 3, 32, 32), x_T=x, timesteps=5)
    img = x
    for i in (3, 2, 1, 0):
        img, _ = s.p_sample_ddim(img, c, torch.full((2,), int(s.ddim_timesteps[i]), device="cuda", dtype=torch.long), index=i)
    assert torch.equal(sub, img)


def _run3(fr, eta, scale, noise=None):
    from dsml_thesis_amd.ddim import DDIMSampler
    c, uc = _cond(fr)
    s = DDIMSampler(fr)
    s.make_schedule(200, ddim_eta=eta, verbose=False)
    img = rnd(51, 2, 3, 32, 32).cuda()
    for i, step in enumerate(np.flip(s.ddim_timesteps)[:3]):
        t = torch.full((2,), int(step), device="cuda", dtype=torch.long)
        img, _ = s.p_sample_ddim(img, c, t, index=200 - i - 1, unconditional_guidance_scale=scale,
                                 unconditional_conditioning=uc if scale != 1.0 else None,
                                 noise=None if noise is None else noise[i].cuda())
    return img


def test_ddim_three_steps_cfg_and_eta(fr):
    g = golden("g5_sampling_fr.npz")
    close(_run3(fr, 0.0, 1.0), g["s200_e0_cfg1"], 1e-4, 1e-4)
    close(_run3(fr, 0.0, 3.0), g["s200_e0_cfg3"], 1e-4, 1e-4)
    close(_run3(fr, 1.0, 1.0, T(g["s200_e1_noise"])), g["s200_e1_cfg1"], 1e-4, 1e-4)


def test_ddim_sample_cfg_loop_matches_stepwise(fr):
    from dsml_thesis_amd.ddim import DDIMSampler
    c, uc = _cond(fr)
    xT = rnd(51, 2, 3, 32, 32).cuda()
    s = DDIMSampler(fr)
    out, _ = s.sample(S=200, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False,
                      unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
    assert torch.isfinite(out).all()
    # first three steps of the fused loop == the stepwise API (same kernels, same order)
    ref3 = _run3(fr, 0.0, 3.0)
    s3 = DDIMSampler(fr)
    # run the fused loop for exactly 3 steps by stopping through the callback
    class Stop(Exception):
        pass
    seen = {}
    def cb(i):
        if i == 2:
            seen["x"] = s3.model.model.diffusion_model.program(4, 32, 32, 1, 0).inputs["x"][:2].clone()
            raise Stop()
    with pytest.raises(Stop):
        s3.sample(S=200, batch_size=2, shape=[3, 32, 32], conditioning=c, eta=0.0, x_T=xT, verbose=False,
                  unconditional_guidance_scale=3.0, unconditional_conditioning=uc, callback=cb)
    assert torch.equal(seen["x"], ref3)


class _ShiftCorrector:
    """the deterministic stand-in tools/make_golden.py used for the (unshipped) score-corrector plugin"""

    def modify_score(self, model, e_t, x, t, c, strength=0.1):
        return e_t - strength * x * (t.float().view(-1, 1, 1, 1) / 1000.0)


def _close_but_for_codebook_ties(a, b, tol, max_frac=0.01):
    """After a nearest-codebook snap a latent within ~1e-5 of a cell boundary may legitimately land on the neighbouring
    code: all elements must agree within `tol` except at most `max_frac` of them."""
    a, b = a.float().cpu(), torch.as_tensor(np.asarray(b)).float()
    bad = ((a - b).abs() > tol + tol * b.abs()).float().mean().item()
    assert bad <= max_frac, f"{100 * bad:.2f} % of the elements differ by more than {tol}"


def test_sampler_options_against_reference_fixtures(fr):
    """ddim.py:112-203 / ddpm.py:1049-1216 options that no shipped script sets -- mask + x0 blend, temperature, quantize_x0,
    score_corrector (DDIM); clip_denoised, quantize_denoised, mask (ancestral loop) -- against runs of the reference's own
    samplers with those options (tests/golden/g12_sampler_options.npz; the reference's noise draws are injected)."""
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g12_sampler_options.npz")
    c, uc = _cond(fr)
    xT, x0 = rnd(51, 2, 3, 32, 32).cuda(), (0.5 * rnd(52, 2, 3, 32, 32)).cuda()
    mask = (rnd(53, 2, 1, 32, 32) > 0).float().cuda()
    s = DDIMSampler(fr)
    kw = dict(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, x_T=xT, verbose=False)
    out, _ = s.sample(eta=1.0, mask=mask, x0=x0, temperature=0.7, unconditional_guidance_scale=3.0,
                      unconditional_conditioning=uc, noise=T(g["step_noise"]).cuda(), mask_noise=T(g["mask_noise"]).cuda(), **kw)
    close(out, g["ddim_mask_temp_cfg"], 2e-4, 2e-4)
    out_g, _ = s.sample(eta=1.0, mask=mask, x0=x0, temperature=0.7, unconditional_guidance_scale=3.0, use_graph=True,
                        unconditional_conditioning=uc, noise=T(g["step_noise"]).cuda(), mask_noise=T(g["mask_noise"]).cuda(), **kw)
    assert torch.equal(out, out_g), "the option steps are device-side work: hipGraph replay == eager"
    out, inter = s.sample(eta=0.0, quantize_x0=True, log_every_t=1, **kw)
    _close_but_for_codebook_ties(out, g["ddim_quantize"], 2e-4)
    _close_but_for_codebook_ties(inter["pred_x0"][-1], g["ddim_quantize_pred_x0"], 2e-4)
    out, _ = s.sample(eta=0.0, score_corrector=_ShiftCorrector(), corrector_kwargs=dict(strength=0.2),
                      unconditional_guidance_scale=3.0, unconditional_conditioning=uc, **kw)
    close(out, g["ddim_corrector_cfg"], 2e-4, 2e-4)
    fr.clip_denoised = True
    try:
        out = fr.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=3, verbose=False, quantize_denoised=True, mask=mask, x0=x0,
                               noise=list(T(g["ddpm_noise"]).cuda()), mask_noise=list(T(g["ddpm_mask_noise"]).cuda()))
    finally:
        fr.clip_denoised = False
    _close_but_for_codebook_ties(out, g["ddpm_clip_quant_mask"], 2e-4)
    # noise_dropout has no cross-device fixture (its Bernoulli draw is the device's): a fraction p of the step noise is
    # zeroed and the rest scaled by 1/(1-p) (F.dropout), so the run stays finite and differs from the p = 0 run
    a, _ = s.sample(eta=1.0, noise=T(g["step_noise"]).cuda(), **kw)
    torch.manual_seed(3)
    b_, _ = s.sample(eta=1.0, noise=T(g["step_noise"]).cuda(), noise_dropout=0.5, **kw)
    assert torch.isfinite(b_).all() and not torch.equal(a, b_)


def test_p_sample_loop_T3(fr):
    g = golden("g5_sampling_fr.npz")
    c, _ = _cond(fr)
    out = fr.p_sample_loop(c, (2, 3, 32, 32), x_T=rnd(51, 2, 3, 32, 32).cuda(), timesteps=3, verbose=False,
                           noise=list(T(g["p_sample_loop_noise"]).cuda()))
    close(out, g["p_sample_loop_T3"], 1e-4, 1e-4)


def test_first_stage_winograd_route_golden(monkeypatch):
    """The VQGAN's 512-channel ResnetBlocks (model.py:95-129) take the Winograd route only for large batches; force it for
    one frame and hold decoder and encoder to the same reference fixtures and bounds as the direct route."""
    from dsml_thesis_amd.engine import NetBuilder
    monkeypatch.setattr(NetBuilder, "WINO_MIN_TILES", 1)
    monkeypatch.setattr(NetBuilder, "UP_MIN_PIXELS", 1)
    m = make_fr_model()
    z = rnd(61, 1, 3, 32, 32).cuda()
    img, idx = m.first_stage_model.decode(z, return_indices=True)
    fs = m.first_stage_model
    launches = [c[3] for pg in fs._programs.values() for c in pg.calls]
    assert sum(launches.count(k) for k in ("ldmk_winograd_input", "ldmk_winograd_input_ps", "ldmk_winograd_input_ps_h2")) >= 10
    assert sum(launches.count(k) for k in ("ldmk_upconv_gather", "ldmk_upconv_gather_ps", "ldmk_upconv_gather_ps_h2")) == 1    # 512 -> 512, 32 -> 64
    assert np.array_equal(idx.cpu().numpy(), golden("g6_vqgan.npz")["vq_idx"].reshape(-1))
    close(img, golden("g11_northstar.npz")["decoded128"], 1e-4, 1e-4)
    x = torch.tanh(rnd(64, 1, 3, 128, 128)).cuda()
    close(m.encode_first_stage(x), golden("g6_vqgan.npz")["encoded"], 1e-4, 1e-4)


def test_first_stage_split_arithmetic_golden(monkeypatch):
    """VQGAN decoder and encoder with EVERY eligible GEMM (ResnetBlock convolutions, Winograd planes, upsampling phases, the
    AttnBlock's 1x1 projections) in the fp32-accurate bf16x3 arithmetic -- the clip / decode benchmarks run the large ones in
    it -- against the same reference fixtures and bounds as the f32 matrix-core form; the codebook indices stay bit-exact."""
    from dsml_thesis_amd import engine, lib as L
    from dsml_thesis_amd.engine import NetBuilder
    monkeypatch.setattr(engine, "x3_plan", lambda a, m, far=False: (1, 1) if a.epi == L.EPI_GEGLU else (5, 1))
    for wino in (False, True):
        if wino:
            monkeypatch.setattr(NetBuilder, "WINO_MIN_TILES", 1)
            monkeypatch.setattr(NetBuilder, "UP_MIN_PIXELS", 1)
        m = make_fr_model()
        z = rnd(61, 1, 3, 32, 32).cuda()
        img, idx = m.first_stage_model.decode(z, return_indices=True)
        fs = m.first_stage_model
        gemms = [c[2] for pg in fs._programs.values() for c in pg.calls if c[3] == "ldmk_igemm"]
        n3 = sum(1 for a in gemms if a.compute in (L.COMPUTE_BF16X3, L.COMPUTE_F16X2))
        assert n3 >= 0.7 * len(gemms), (n3, len(gemms))
        assert np.array_equal(idx.cpu().numpy(), golden("g6_vqgan.npz")["vq_idx"].reshape(-1))
        close(img, golden("g11_northstar.npz")["decoded128"], 1e-4, 1e-4)
        x = torch.tanh(rnd(64, 1, 3, 128, 128)).cuda()
        close(m.encode_first_stage(x), golden("g6_vqgan.npz")["encoded"], 1e-4, 1e-4)


def test_first_stage_decode_encode_golden():
    g = golden("g6_vqgan.npz")
    m = make_fr_model()
    z = rnd(61, 1, 3, 32, 32).cuda()
    img, idx = m.first_stage_model.decode(z, return_indices=True)
    assert np.array_equal(idx.cpu().numpy(), g["vq_idx"].reshape(-1))          # index work: bit-exact
    close(m.decode_first_stage(z), g["decoded"].astype(np.float32), 2e-3, 2e-3)      # g6 stores fp16
    close(img, golden("g11_northstar.npz")["decoded128"], 1e-4, 1e-4)                 # the same frame in fp32
    st = g["decoded_stats"]
    assert abs(img.abs().max().item() - st[0]) < 2e-3 and abs(img.std().item() - st[2]) < 1e-4
    # tight check against the oracle recomputed here in fp32
    sd = W.synth_state_dict(W.vqmodel_param_shapes(W.VQ_F4))
    ref, ridx = O.decode_first_stage(sd, W.VQ_F4, z.cpu())
    assert torch.equal(idx.cpu().long(), ridx.reshape(-1))
    close(img, ref, 1e-4, 1e-4)
    x = torch.tanh(rnd(64, 1, 3, 128, 128)).cuda()
    close(m.encode_first_stage(x), g["encoded"], 1e-4, 1e-4)
    # batch > 1 and no-quantise path
    z2 = rnd(65, 3, 3, 32, 32).cuda()
    a = m.first_stage_model.decode(z2, force_not_quantize=True)
    b = torch.cat([m.first_stage_model.decode(z2[i:i + 1], force_not_quantize=True) for i in range(3)])
    close(a, b.cpu(), 1e-4, 1e-4)        # tile shapes (K-summation order) differ between batch 3 and batch 1
    from dsml_thesis_amd import ops
    frames = ops.postprocess_frames(a)
    assert frames.shape == (3, 128, 128, 3) and frames.min() >= 0 and frames.max() <= 1


def test_talking_face_progressive_golden():
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g7_talking_face.npz")
    m = make_tf_model(gain=0.25, seq_len=3)
    close(m.cond_stage_model_2(rnd(74, 2, 3, 768).cuda()), g["audio_att"], 1e-4, 1e-5)
    Tn, S = 3, 4
    audio = rnd(75, Tn, 768).cuda()
    masked = torch.tanh(rnd(76, Tn, 3, 128, 128))
    masked[:, :, 70:, :] = -1.0
    ident = torch.tanh(rnd(77, 1, 3, 128, 128)).cuda()
    c1 = m.cond_stage_model_1.embedding(torch.tensor([[4]], device="cuda"))
    xid = m.encode_first_stage(ident)
    close(xid, g["xid"], 1e-4, 1e-4)
    xT = rnd(78, Tn, 1, 3, 32, 32).cuda()
    s = DDIMSampler(m)
    for fixed, tag in ((False, "autoreg"), (True, "fixed")):
        frames, _ = s.progressive_sampling(c1, xid, masked.cuda(), audio, S, 1, Tn, [3, 32, 32], 1, eta=0.0, x_T=xT,
                                           fixed_identity=fixed, verbose=False)
        close(torch.cat(frames), g[f"frames_{tag}"], 2e-4, 2e-4)
    # the batched TF program with the plan set of the 128-frame clip (configs[2]/[3] run fixed-identity frames at
    # policy_batch = 128: bf16x3 table shapes, Winograd at every level) against the same reference frames
    frames, _ = s.progressive_sampling(c1, xid, masked.cuda(), audio, S, 1, Tn, [3, 32, 32], 1, eta=0.0, x_T=xT,
                                       fixed_identity=True, verbose=False, policy_batch=128)
    unet = m.model.diffusion_model
    assert unet.policy_batch == 128 and not getattr(unet.program(Tn, 32, 32, 1, 6), "small_route", False)
    close(torch.cat(frames), g["frames_fixed"], 2e-4, 2e-4)
    # eager launches == captured graph, frame chain included
    fr_e, _ = s.progressive_sampling(c1, xid, masked.cuda(), audio, S, 1, Tn, [3, 32, 32], 1, eta=0.0, x_T=xT,
                                     use_graph=False, verbose=False)
    fr_g, _ = s.progressive_sampling(c1, xid, masked.cuda(), audio, S, 1, Tn, [3, 32, 32], 1, eta=0.0, x_T=xT,
                                     use_graph=True, verbose=False)
    assert torch.equal(torch.cat(fr_e), torch.cat(fr_g))


@pytest.mark.parametrize("policy", [16, None])
def test_talking_face_clips_in_lock_step(policy):
    """progressive_sampling(..., clips=V): V independent videos (the reference loops over 150 of them,
    progressive_sampling_difftalk.py:336), each its own autoregressive chain, advanced frame by frame as ONE batch.  Clip 0
    is the g7 fixture's clip (reference frames); lengths are ragged (3, 2, 3 frames: the short clip leaves the batch); every
    clip's frames equal bit for bit those of progressive_sampling on that clip alone at the same plan policy."""
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g7_talking_face.npz")
    m = make_tf_model(gain=0.25, seq_len=3)
    S, lens = 4, [3, 2, 3]
    V = len(lens)
    audio = [rnd(75 + 100 * v, T_, 768).cuda() for v, T_ in enumerate(lens)]
    masked = []
    for v, T_ in enumerate(lens):
        mk = torch.tanh(rnd(76 + 100 * v, T_, 3, 128, 128))
        mk[:, :, 70:, :] = -1.0
        masked.append(mk.cuda())
    ident = torch.cat([torch.tanh(rnd(77 + 100 * v, 1, 3, 128, 128)) for v in range(V)]).cuda()
    labels = torch.tensor([[4], [1], [7]], device="cuda")
    c1 = m.cond_stage_model_1.embedding(labels)                               # (V,1,256)
    xid = torch.cat([m.encode_first_stage(ident[v:v + 1]) for v in range(V)])
    xT = [rnd(78 + 100 * v, T_, 1, 3, 32, 32).cuda() for v, T_ in enumerate(lens)]
    s = DDIMSampler(m)
    clips, _ = s.progressive_sampling(c1, xid, masked, audio, S, 1, None, [3, 32, 32], 1, eta=0.0, x_T=xT, clips=V,
                                      policy_batch=policy, verbose=False)
    assert [len(f) for f in clips] == lens
    close(torch.cat(clips[0]), g["frames_autoreg"], 2e-4, 2e-4)
    for v in range(V):
        alone, _ = s.progressive_sampling(c1[v:v + 1], xid[v:v + 1], masked[v], audio[v], S, 1, lens[v], [3, 32, 32], 1,
                                          eta=0.0, x_T=xT[v], policy_batch=V if policy is None else policy, verbose=False)
        assert torch.equal(torch.cat(alone), torch.cat(clips[v])), f"clip {v}"
    # eager launches == hipGraph replay (one captured step per active count)
    eager, _ = s.progressive_sampling(c1, xid, masked, audio, S, 1, None, [3, 32, 32], 1, eta=0.0, x_T=xT, clips=V,
                                      policy_batch=policy, use_graph=False, verbose=False)
    assert all(torch.equal(torch.cat(a), torch.cat(b)) for a, b in zip(eager, clips))


def test_ema_scope_swaps_weights_and_repacks(fr):
    """Inside ema_scope EVERY weight the kernels read is the EMA one (packed conv / linear matrices included, not only
    the aliased biases): the in-scope output equals the oracle evaluated on the shadow weights."""
    c, _ = _cond(fr)
    x, t = rnd(90, 1, 3, 32, 32).cuda(), torch.tensor([400], device="cuda")
    base = fr.apply_model(x, t, c[:1])
    # a shadow that differs from the live weights in every tensor (the constructor's shadow holds the zero-init sites)
    ema_sd = W.synth_state_dict(W.unet_param_shapes(W.FR_UNET), seed=5)
    with torch.no_grad():
        for k, v in ema_sd.items():
            fr.model_ema.shadow_of("diffusion_model." + k).copy_(v)
        ref = O.unet_forward(ema_sd, W.FR_UNET, x.cpu(), t.cpu(), c[:1].detach().cpu())
    with fr.ema_scope():
        ema = fr.apply_model(x, t, c[:1])
    again = fr.apply_model(x, t, c[:1])
    close(ema, ref, 1.5e-4, 1.5e-4)
    assert not torch.allclose(base, ema) and torch.equal(base, again)
    # the batched program (job batch 16): inside the scope the bf16x3 weight images and the Winograd / phase planes must be
    # the EMA weights' too (ops.pack_wsplit follows the re-pack), and the live ones again after it
    unet = fr.model.diffusion_model
    unet.policy_batch = 16
    try:
        base16 = fr.apply_model(x, t, c[:1])
        pg = unet.program(1, 32, 32, 1, 0)
        assert not getattr(pg, "small_route", False)
        from dsml_thesis_amd import lib as L
        assert sum(1 for cl in pg.calls if cl[3] == "ldmk_igemm" and cl[2].compute in (L.COMPUTE_BF16X3, L.COMPUTE_F16X2)) >= 40
        # the second scope re-packs NOTHING: both weight sets (kernel layouts, programs, graphs) are kept under their names and
        # swapped by reference (UNetModel.adopt_weights) -- the shadow and the stored weights have not changed since
        packs = []
        orig = unet.pack_weights
        unet.pack_weights = lambda: (packs.append(1), orig())[1]
        try:
            with fr.ema_scope():
                ema16 = fr.apply_model(x, t, c[:1])
            again16 = fr.apply_model(x, t, c[:1])
        finally:
            del unet.pack_weights
        assert not packs, f"{len(packs)} re-packs in a repeated ema_scope"
    finally:
        unet.policy_batch = None
    close(ema16, ref, 1.5e-4, 1.5e-4)
    close(base16, base.cpu(), 3e-5, 3e-5)
    assert not torch.allclose(base16, ema16) and torch.equal(base16, again16)
    keys = fr.state_dict().keys()
    assert "model.diffusion_model.input_blocks.1.0.in_layers.2.weight" in keys
    assert "model_ema.diffusion_modelinput_blocks10in_layers2weight" in keys
    assert "first_stage_model.decoder.up.2.attn.1.q.weight" in keys and "cond_stage_model.embedding.weight" in keys


def test_northstar_trajectory_and_decode_golden():
    """BASELINE.json's metric shape, against outputs of the real reference (tests/golden/g11_northstar.npz):
    DDIMSampler.sample S=4 at 64x64x4 (B=2), then decode_first_stage 4x64x64 -> 3x256x256 with bit-exact indices."""
    from dsml_thesis_amd import synth
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g11_northstar.npz")
    m = make_fr_model(gain=0.25, unet=synth.NS_UNET, vq=synth.VQ_F4_256)
    labels = torch.tensor([3, 4], device="cuda")
    c = m.cond_stage_model.embedding(labels[:, None])
    xT = rnd(111, 2, 4, 64, 64).cuda()
    for use_graph in (False, True):
        out, inter = DDIMSampler(m).sample(S=4, batch_size=2, shape=[4, 64, 64], conditioning=c, eta=0.0, x_T=xT,
                                           verbose=False, log_every_t=1, use_graph=use_graph)
        close(inter["x_inter"][1], g["x_inter_1"], 2e-4, 2e-4)
        close(out, g["sample_S4"], 1.5e-4, 1.5e-4)
    z = rnd(112, 1, 4, 64, 64).cuda()
    img, idx = m.first_stage_model.decode(z, return_indices=True)
    assert np.array_equal(idx.cpu().numpy(), g["vq4_idx"].reshape(-1))
    close(img, g["decoded256"], 1e-4, 1e-4)
    close(m.decode_first_stage(z, force_not_quantize=True), g["decoded256_noquant"], 1e-4, 1e-4)


def test_northstar_trajectory_in_the_split_arithmetic(monkeypatch):
    """The same S = 4 trajectory at 64x64x4 with the batched program (job batch 16) and EVERY eligible GEMM of the UNet forced
    into the bf16x3 arithmetic, eager and hipGraph: held to the reference's trajectory with the unchanged bounds, and the two
    launch modes equal bit for bit."""
    from dsml_thesis_amd import engine, lib as L, synth
    from dsml_thesis_amd.ddim import DDIMSampler
    monkeypatch.setattr(engine, "x3_plan", lambda a, m, far=False: (1, 1) if a.epi == L.EPI_GEGLU else (5, 1))
    g = golden("g11_northstar.npz")
    m = make_fr_model(gain=0.25, unet=synth.NS_UNET, vq=synth.VQ_F4_256)
    m.model.diffusion_model.policy_batch = 16
    labels = torch.tensor([3, 4], device="cuda")
    c = m.cond_stage_model.embedding(labels[:, None])
    xT = rnd(111, 2, 4, 64, 64).cuda()
    outs = []
    for use_graph in (False, True):
        out, inter = DDIMSampler(m).sample(S=4, batch_size=2, shape=[4, 64, 64], conditioning=c, eta=0.0, x_T=xT,
                                           verbose=False, log_every_t=1, use_graph=use_graph)
        close(inter["x_inter"][1], g["x_inter_1"], 2e-4, 2e-4)
        close(out, g["sample_S4"], 1.5e-4, 1.5e-4)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    pgs = [pg for pg in m.model.diffusion_model._programs.values()]
    gemms = [c_[2] for pg in pgs for c_ in pg.calls if c_[3] == "ldmk_igemm"]
    assert gemms and sum(1 for a in gemms if a.compute in (L.COMPUTE_BF16X3, L.COMPUTE_F16X2)) >= len(gemms) - 8
    assert any(c_[3] in ("ldmk_attn_self_x3", "ldmk_attn_self_x3p", "ldmk_attn_self_h2", "ldmk_attn_self_h2_ps", "ldmk_attn_self_h2_tiles") for pg in pgs for c_ in pg.calls)


def test_sharded_sampling_bitwise_equals_single_gpu(fr):
    """Config 4's invariant on one GPU: the per-rank blocks of an 8-way / 3-way sharded job, computed one after
    the other with the job-wide tile policy, concatenate to exactly the single-GPU result (latents and frames)."""
    from dsml_thesis_amd.ddim import DDIMSampler
    from dsml_thesis_amd.parallel import sample_sharded, shard_range
    n_items, S = 8, 4          # S must divide 1000 like in the reference
    s = DDIMSampler(fr)
    labels = torch.arange(n_items, device="cuda") % 8
    cond = lambda lo, hi: fr.cond_stage_model.embedding(labels[lo:hi, None])
    full = sample_sharded(s, S, n_items, (3, 32, 32), cond, seed=3, rank=0, world_size=1)
    full_z = sample_sharded(s, S, n_items, (3, 32, 32), cond, seed=3, rank=0, world_size=1, decode=False)
    assert full.shape == (n_items, 128, 128, 3) and float(full.min()) >= 0 and float(full.max()) <= 1
    for world in (8, 3):
        parts, parts_z = [], []
        for r in range(world):
            lo, hi = shard_range(n_items, world, r)
            if hi > lo:
                # what rank r computes (the all-gather itself is covered by the gloo test / RCCL on the node)
                kw = dict(seed=3, rank=0, world_size=1, _noise_offset=lo, _policy_items=n_items)
                sub = lambda a, b, lo=lo: cond(lo + a, lo + b)
                parts_z.append(sample_sharded(s, S, hi - lo, (3, 32, 32), sub, decode=False, **kw))
                parts.append(sample_sharded(s, S, hi - lo, (3, 32, 32), sub, **kw))
        assert torch.equal(torch.cat(parts_z), full_z), f"latents, world={world}"
        assert torch.equal(torch.cat(parts), full), f"frames, world={world}"
        # policy="shard": every rank plans for its own block (here 1 or 3 items: other tiles, the small-batch route) -- the
        # same items within the tolerance of a short trajectory (other summation orders only), what `bench.py --gpus N` times
        parts_s = [sample_sharded(s, S, n_items, (3, 32, 32), cond, seed=3, rank=r, world_size=world, decode=False, policy="shard")
                   for r in range(world) if shard_range(n_items, world, r)[1] > shard_range(n_items, world, r)[0]]
        close(torch.cat(parts_s), full_z.cpu(), 1.5e-4, 1.5e-4)


def test_ddim_inversion_and_tuned_sampling_golden(fr):
    """forward DDIM (inversion) + regeneration on the strength-scaled schedule, against the
    reference's compute_latents.DDIMSampler (fixture g8), eager and hipGraph."""
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g8_inversion.npz")
    c, uc = _cond(fr)
    x0 = rnd(81, 2, 3, 32, 32).cuda()
    s = DDIMSampler(fr)
    for tag, scale in (("cfg1", 1.0), ("cfg3", 3.0)):
        kw = dict(unconditional_guidance_scale=scale, unconditional_conditioning=uc if scale != 1.0 else None)
        img, lat, _ = s.compute_latents(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, x0=x0, strength=0.5,
                                        verbose=False, **kw)
        assert np.array_equal(s.ddim_timesteps, g["timesteps"])
        close(lat, g[f"xlat_{tag}"], 1e-4, 1e-4)
        close(img, g[f"img_{tag}"], 1.5e-4, 1.5e-4)
        img_g, lat_g, _ = s.compute_latents(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c, x0=x0, strength=0.5,
                                            verbose=False, use_graph=True, **kw)
        assert torch.equal(lat, lat_g) and torch.equal(img, img_g)
        # latent_manipulation_tuned.ddim_tuned_sampling == the reverse half started from the stored latent
        again = s.ddim_tuned_sampling(4, 2, [3, 32, 32], lat, c, strength=0.5, verbose=False, **kw)
        assert torch.equal(again, img)


def test_first_stage_non_square_and_batch_vs_oracle():
    m = make_fr_model()
    sd = W.synth_state_dict(W.vqmodel_param_shapes(W.VQ_F4))
    z = rnd(95, 2, 3, 8, 12)
    ref, _ = O.decode_first_stage(sd, W.VQ_F4, z, force_not_quantize=True)
    close(m.first_stage_model.decode(z.cuda(), force_not_quantize=True), ref, 1e-4, 1e-4)
    img = torch.tanh(rnd(96, 2, 3, 32, 48))
    close(m.encode_first_stage(img.cuda()), O.encode_first_stage(sd, W.VQ_F4, img), 1e-4, 1e-4)


def test_talking_face_front_end_kernels():
    """the audio window encoder as one fused kernel (window 17 = the shipped config) and the
    lower-face mask, against the oracle restatement / the reference's slicing semantics."""
    from dsml_thesis_amd.encoders import Conv1DTemporalAttention, mask_lower_face_
    m = Conv1DTemporalAttention(17).cuda()
    sd = W.synth_state_dict(W.audio_attention_param_shapes(17))
    m.load_state_dict(sd)
    x = rnd(97, 5, 17, 768)
    close(m(x.cuda()), O.audio_temporal_attention(sd, x), 1e-4, 1e-5)
    img = rnd(98, 3, 3, 16, 12)
    y0 = [4, 0, 16]
    ref = img.clone()
    for i, y in enumerate(y0):
        ref[i, :, y:, :] = -1.0            # masked_img[min_y:, :, :] = -1 (HWC in the reference, CHW here)
    out = mask_lower_face_(img.cuda(), y0)
    assert torch.equal(out.cpu(), ref)


def test_p_sample_loop_graph_and_full_length(fr):
    """The ancestral loop as a device-resident program: hipGraph replay is statistically the same process (noise is
    drawn inside the captured step) and the full 1000-step chain stays finite; eager with injected noise is pinned by
    the golden above."""
    c, _ = _cond(fr)
    xT = rnd(51, 2, 3, 32, 32).cuda()
    torch.manual_seed(3)
    a, inter = fr.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=20, return_intermediates=True, log_every_t=5)
    assert len(inter) >= 5 and torch.isfinite(a).all()
    g1 = fr.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=20, use_graph=True)
    g2 = fr.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=20, use_graph=True)
    assert torch.isfinite(g1).all() and not torch.equal(g1, g2)          # fresh noise on every replay
    zero = [torch.zeros(2, 3, 32, 32, device="cuda")] * 20
    d1 = fr.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=20, noise=zero)
    d2 = fr.p_sample_loop(c, (2, 3, 32, 32), x_T=xT, timesteps=20, noise=zero)
    assert torch.equal(d1, d2)
    out = fr.sample(c, batch_size=2, x_T=xT, use_graph=True)              # LatentDiffusion.sample -> 1000 steps
    assert out.shape == (2, 3, 32, 32) and torch.isfinite(out).all()


def test_class_conditional_model_through_the_samplers():
    """conditioning_key 'adm' under the samplers (ddpm.py:1417-1419: the conditioning IS the class-label vector y): a real-surface
    LatentDiffusion around the class-conditional UNet (use_scale_shift_norm, resblock_updown), DDIMSampler.sample plain and with
    classifier-free guidance (labels doubled as [uncond | cond], ddim.py:175) -- eager == hipGraph bitwise -- and p_sample_loop,
    against the REAL reference's outputs (g15: LatentDiffusion(conditioning_key='adm') + DDIMSampler from /root/reference).  The
    label-embedding rows are written once per run into the launch program's `y_emb` input; through round 5's first session only
    apply_model / p_sample_ddim knew the key and the loops failed on the 1-D labels."""
    from dsml_thesis_amd.ddim import DDIMSampler
    from dsml_thesis_amd.ddpm import LatentDiffusion
    from dsml_thesis_amd.synth import fr_config, load_recipe
    g = golden("g15_updown.npz")
    cfg = fr_config(unet=W.UPDOWN_ADM_UNET)
    cfg.update(conditioning_key="adm")
    m = LatentDiffusion(**cfg)
    load_recipe(m.model.diffusion_model, gain=0.25)
    m = m.cuda().eval()
    assert m.model.conditioning_key == "adm"
    xT, y, uy = rnd(176, 2, 3, 16, 16).cuda(), torch.tensor([7, 2]).cuda(), torch.tensor([0, 0]).cuda()
    kw = dict(S=4, batch_size=2, shape=[3, 16, 16], conditioning=y, eta=0.0, x_T=xT, verbose=False)
    s = DDIMSampler(m)
    out, _ = s.sample(**kw)
    close(out, g["adm_ddim4"], 1.5e-4, 1.5e-4)
    out_g, _ = s.sample(use_graph=True, **kw)
    assert torch.equal(out, out_g)
    for form in ([y], {"c_crossattn": [y]}):               # the other forms apply_model accepts for the labels (ddpm.py:893-994)
        assert torch.equal(s.sample(**dict(kw, conditioning=form))[0], out)
    assert torch.equal(m.p_sample_loop([y], (2, 3, 16, 16), x_T=xT, timesteps=2, verbose=False, noise=list(T(g["adm_ddpm3_noise"]).cuda())),
                       m.p_sample_loop(y, (2, 3, 16, 16), x_T=xT, timesteps=2, verbose=False, noise=list(T(g["adm_ddpm3_noise"]).cuda())))
    cfg3 = dict(unconditional_guidance_scale=3.0, unconditional_conditioning=uy)
    out3, _ = s.sample(**kw, **cfg3)
    close(out3, g["adm_ddim4_cfg3"], 1.5e-4, 1.5e-4)
    out3g, _ = s.sample(use_graph=True, **kw, **cfg3)
    assert torch.equal(out3, out3g) and not torch.equal(out3, out)
    # one step with the reference's signature goes through apply_model -> DiffusionWrapper('adm')
    s.make_schedule(4, ddim_eta=0.0, verbose=False)
    t3 = torch.full((2,), int(s.ddim_timesteps[3]), device="cuda", dtype=torch.long)
    x1, _ = s.p_sample_ddim(xT, y, t3, 3)
    x1c, _ = s.p_sample_ddim(xT, y, t3, 3, **cfg3)
    assert torch.isfinite(x1).all() and torch.isfinite(x1c).all() and not torch.equal(x1, x1c)
    out = m.p_sample_loop(y, (2, 3, 16, 16), x_T=xT, timesteps=3, verbose=False, noise=list(T(g["adm_ddpm3_noise"]).cuda()))
    close(out, g["adm_ddpm3"], 1e-4, 1e-4)
    with pytest.raises(AssertionError, match="class labels"):
        s.sample(**dict(kw, conditioning=torch.zeros(2, 1, 512).cuda()))


def test_p_mean_variance_with_the_reference_signature(fr):
    """LatentDiffusion.p_mean_variance (ddpm.py:1049-1078) -- one UNet evaluation, then predict_start_from_noise and q_posterior on
    the schedule buffers -- returns what the fused ldmk_ddpm_step of p_sample / p_sample_loop consumes: mean + sigma z equals
    p_sample on the same noise (to rounding: the kernel fuses the arithmetic), and the oracle's update on the oracle's eps."""
    c = _cond(fr)[0].detach()
    x, nz = rnd(60, 2, 3, 32, 32).cuda(), rnd(61, 2, 3, 32, 32).cuda()
    t = torch.tensor([0, 640], device="cuda")
    mean, var, logvar, x0 = fr.p_mean_variance(x, c, t, clip_denoised=False, return_x0=True)
    assert mean.shape == x.shape and var.shape == (2, 1, 1, 1) and x0.shape == x.shape
    nonzero = (1 - (t == 0).float()).view(2, 1, 1, 1)
    step = mean + nonzero * (0.5 * logvar).exp() * nz
    close(fr.p_sample(x, c, t, noise=nz), step.cpu(), 1e-5, 1e-5)
    sd = {k: v.detach().cpu() for k, v in fr.model.diffusion_model.state_dict().items()}
    sched = O.register_schedule(**W.SCHEDULE)
    with torch.no_grad():
        eps = O.unet_forward(sd, W.FR_UNET, x.cpu(), t.cpu(), c.cpu())
    close(step, O.ddpm_update(sched, x.cpu(), eps, t.cpu(), nz.cpu()), 1e-4, 1e-4)
    xc = fr.p_mean_variance(x, c, t, clip_denoised=True, return_x0=True)[3]
    assert float(xc.abs().max()) <= 1.0 and len(fr.p_mean_variance(x, c, t, clip_denoised=False)) == 3


def test_stochastic_encode_and_decode_img2img_pair():
    """DDIMSampler.stochastic_encode / decode (ddim.py:205-219, ddim2cond.py:198-250; the SDEdit-style img2img pair) against the
    REAL talking-face sampler around the real LatentDiffusion (g16): encode on the DDIM subsequence and on the model's own steps
    (one ldmk_q_sample launch, the sampler's float32-root tables), decode = the device-resident DDIM loop started at index
    t_start - 1 -- eta 0 eager == hipGraph bitwise, eta 1 with the reference's seeded draws replayed."""
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g16_sdedit.npz")
    m = make_tf_model(gain=0.25, seq_len=3)
    x0, nz = rnd(180, 2, 3, 32, 32).cuda(), rnd(181, 2, 3, 32, 32).cuda()
    cond = {"class_label_&_audio": rnd(182, 2, 1, 1024).cuda(), "motion_&_id": rnd(183, 2, 6, 32, 32).cuda()}
    s = DDIMSampler(m)
    s.make_schedule(5, ddim_eta=0.0, verbose=False)
    close(s.stochastic_encode(x0, torch.tensor([3, 1]), noise=nz), g["enc"], 1e-6, 1e-6)
    close(s.stochastic_encode(x0, torch.tensor([640, 7]).cuda(), use_original_steps=True, noise=nz), g["enc_orig"], 1e-6, 1e-6)
    torch.manual_seed(0)
    assert not torch.equal(s.stochastic_encode(x0, torch.tensor([3, 1])), s.stochastic_encode(x0, torch.tensor([3, 1])))   # own draws
    with pytest.raises(AssertionError, match="out of the table"):
        s.stochastic_encode(x0, torch.tensor([5, 0]))
    x_lat = s.stochastic_encode(x0, torch.tensor([2, 2]), noise=nz)
    dec = s.decode(x_lat, cond, 3)
    close(dec, g["dec3"], 1.5e-4, 1.5e-4)
    assert torch.equal(dec, s.decode(x_lat, cond, 3, use_graph=True))
    full, _ = s.ddim_sampling(cond, tuple(x_lat.shape), x_T=x_lat)
    assert not torch.equal(full, dec)                          # (all five entries is another trajectory)
    s1 = DDIMSampler(m)
    s1.make_schedule(5, ddim_eta=1.0, verbose=False)
    close(s1.decode(x_lat, cond, 3, noise=list(T(g["dec3_eta1_noise"]).cuda())), g["dec3_eta1"], 1.5e-4, 1.5e-4)


def test_latent_manipulation_source_to_target_label(fr):
    """DDIMSampler.latent_manipulation (face_reenactment/latent_manipulation.py:420-490): inversion under the source
    label's conditioning, regeneration under the target label's, against the driver script's own sampler class around the real
    LatentDiffusion (g17), plain and CFG 3; eager == hipGraph bitwise; target == source is compute_latents."""
    from dsml_thesis_amd.ddim import DDIMSampler
    g = golden("g17_manipulation.npz")
    c_src, uc = _cond(fr, (1, 6))
    c_trg, _ = _cond(fr, (3, 0))
    c_src, c_trg, uc = c_src.detach(), c_trg.detach(), uc.detach()
    close(c_src, g["c_src"], 0, 0)
    close(c_trg, g["c_trg"], 0, 0)
    x0 = rnd(190, 2, 3, 32, 32).cuda()
    s = DDIMSampler(fr)
    for tag, scale in (("cfg1", 1.0), ("cfg3", 3.0)):
        kw = dict(S=4, batch_size=2, shape=[3, 32, 32], x0=x0, strength=0.5, verbose=False, unconditional_guidance_scale=scale,
                  unconditional_conditioning=uc if scale != 1.0 else None)
        img, lat, x0_ = s.latent_manipulation(c_src, c_trg, **kw)
        assert x0_ is x0
        close(lat, g[f"xlat_{tag}"], 1e-4, 1e-4)
        close(img, g[f"img_{tag}"], 1.5e-4, 1.5e-4)
        img_g, lat_g, _ = s.latent_manipulation(c_src, c_trg, use_graph=True, **kw)
        assert torch.equal(lat, lat_g) and torch.equal(img, img_g)
    same, lat_s, _ = s.latent_manipulation(c_src, c_src, **kw)
    ref, lat_r, _ = s.compute_latents(S=4, batch_size=2, shape=[3, 32, 32], conditioning=c_src, x0=x0, strength=0.5, verbose=False,
                                      unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
    assert torch.equal(same, ref) and torch.equal(lat_s, lat_r) and not torch.equal(same, img)
    with pytest.raises(AssertionError):
        s.latent_manipulation(c_src, c_trg, **dict(kw, eta=1.0))
