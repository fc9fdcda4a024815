"""CPU: the oracle restatement (oracle/ldm_oracle.py) against the fixtures produced by the REAL
reference (tools/make_golden.py).  Inputs/weights are regenerated from seeds exactly as the
generator did; only the reference's outputs are stored."""
import numpy as np
import pytest
import torch

from conftest import golden, rnd
from oracle import ldm_oracle as O
from oracle import weights as W

torch.set_grad_enabled(False)
T = lambda a: torch.from_numpy(np.asarray(a))


def close(mine, ref, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(torch.as_tensor(mine).float(), torch.as_tensor(ref).float(), rtol=rtol, atol=atol)


def recipe(shapes, seed=0, gain=1.0):
    return W.synth_state_dict(shapes, seed=seed, gain=gain)


def test_g1_schedules():
    g = golden("g1_schedules.npz")
    s = O.register_schedule(**W.SCHEDULE)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
              "posterior_log_variance_clipped", "sqrt_one_minus_alphas_cumprod"):
        assert np.array_equal(s[k].numpy(), g[k]), k
    for S in (50, 200):
        ts = O.make_ddim_timesteps(S)
        assert np.array_equal(ts, g[f"S{S}_timesteps"])
        for eta in (0.0, 1.0):
            tab = O.make_ddim_tables(s["alphas_cumprod"], ts, eta)
            for k, v in tab.items():
                assert np.array_equal(v, g[f"S{S}_eta{int(eta)}_{k}"]), (S, eta, k)
    # headline values quoted in SURVEY.md §8(a) A2
    tab = O.make_ddim_tables(s["alphas_cumprod"], O.make_ddim_timesteps(200), 1.0)
    assert abs(tab["a_t"][0] - 0.9969941) < 1e-6 and abs(tab["sigma_t"][-1] - 0.31256) < 1e-4


def test_g2_timestep_embedding():
    g = golden("g2_timestep_embedding.npz")
    assert np.array_equal(O.timestep_embedding(T(g["t"]), 160).numpy(), g["emb"])


def test_g3_ops():
    g = golden("g3_ops.npz")
    for tag, shape, seed in (("gn160", (2, 160, 8, 8), 11), ("gn480", (1, 480, 4, 4), 12)):
        sd = recipe({"weight": (shape[1],), "bias": (shape[1],)}, seed=1)
        x = rnd(seed, *shape) * 1.5 + 0.3
        close(O.gn_silu(x, sd["weight"], sd["bias"]), g[tag])
    sd = recipe({"weight": (320, 160, 3, 3), "bias": (320,)}, seed=2)
    close(torch.nn.functional.conv2d(rnd(13, 1, 160, 8, 8), sd["weight"], sd["bias"], padding=1), g["conv3x3"])
    keys = {}
    W._resblock(keys, "", 160, 320, 640)
    close(O.resblock(recipe(keys, seed=3), "", rnd(14, 2, 160, 8, 8), rnd(15, 2, 640)), g["resblock"], 1e-5, 2e-5)
    x = rnd(16, 1, 64, 160)
    ca = {"to_q.weight": (160, 160), "to_k.weight": (160, 160), "to_v.weight": (160, 160),
          "to_out.0.weight": (160, 160), "to_out.0.bias": (160,)}
    close(O.cross_attention(recipe(ca, seed=4), "", x, None, 5), g["attn_self"])
    ca["to_k.weight"] = ca["to_v.weight"] = (160, 512)
    for L in (1, 3):
        close(O.cross_attention(recipe(ca, seed=5), "", x, rnd(17 + L, 1, L, 512), 5), g[f"attn_cross_L{L}"])
    ff = {"net.0.proj.weight": (1280, 160), "net.0.proj.bias": (1280,), "net.2.weight": (160, 640),
          "net.2.bias": (160,)}
    close(O.geglu_ff(recipe(ff, seed=6), "", x), g["geglu_ff"])
    keys = {}
    W._spatial_transformer(keys, "", 160, 5, 32, 1, 512)
    sd = recipe(keys, seed=7)
    x4 = rnd(21, 2, 160, 8, 8)
    close(O.spatial_transformer(sd, "", x4, rnd(22, 2, 1, 512), 5), g["spatial_transformer"], 1e-5, 2e-5)
    close(O.spatial_transformer(sd, "", x4, rnd(23, 2, 3, 512), 5), g["spatial_transformer_L3"], 1e-5, 2e-5)
    sd = recipe({"op.weight": (160, 160, 3, 3), "op.bias": (160,)}, seed=8)
    close(torch.nn.functional.conv2d(x4, sd["op.weight"], sd["op.bias"], stride=2, padding=1), g["downsample"])
    sd = recipe({"conv.weight": (160, 160, 3, 3), "conv.bias": (160,)}, seed=9)
    up = torch.nn.functional.interpolate(x4, scale_factor=2, mode="nearest")
    close(torch.nn.functional.conv2d(up, sd["conv.weight"], sd["conv.bias"], padding=1), g["upsample"])
    s = O.register_schedule(**W.SCHEDULE)
    tab = O.make_ddim_tables(s["alphas_cumprod"], O.make_ddim_timesteps(200), 1.0)
    x, e = rnd(31, 2, 3, 32, 32), rnd(32, 2, 3, 32, 32)
    xp, px0 = O.ddim_update(x, e, tab["a_t"][100], tab["a_prev"][100], tab["sigma_t"][100],
                            tab["sqrt_one_minus_at"][100], T(g["ddim_noise"]))
    close(xp, g["ddim_x_prev"], 1e-6, 1e-6)
    close(px0, g["ddim_pred_x0"], 1e-6, 1e-6)
    close(O.ddpm_update(s, x, e, torch.tensor([0, 700]), T(g["ddpm_noise"])), g["ddpm_x_prev"], 1e-6, 1e-6)


def test_g4_unet_fr():
    g = golden("g4_unet_fr.npz")
    sd = recipe(W.unet_param_shapes(W.FR_UNET))
    assert sum(int(np.prod(s)) for s in W.unet_param_shapes(W.FR_UNET).values()) == 156_760_483  # 156.76 M
    out = O.unet_forward(sd, W.FR_UNET, rnd(41, 2, 3, 32, 32), torch.tensor([3, 981]), rnd(42, 2, 1, 512))
    close(out, g["fr_eps"], 1e-4, 1e-4)


def test_g4_unet_northstar_64():
    g = golden("g4_unet_fr.npz")
    sd = recipe(W.unet_param_shapes(W.NS_UNET))
    out = O.unet_forward(sd, W.NS_UNET, rnd(43, 1, 4, 64, 64), torch.tensor([501]), rnd(44, 1, 1, 512))
    close(out, g["ns_eps"], 1e-4, 1e-4)


def _fr_cond():
    emb = W.synth_tensor("embedding.weight", (8, 512))
    unc = W.synth_tensor("uncond_embedding.weight", (1, 512))
    labels = [1, 6]
    return T(emb[labels][:, None]), T(unc[[0, 0]][:, None])


def test_g5_sampling_fr():
    g = golden("g5_sampling_fr.npz")
    sd = recipe(W.unet_param_shapes(W.FR_UNET), gain=0.25)
    s = O.register_schedule(**W.SCHEDULE)
    c, uc = _fr_cond()
    xT = rnd(51, 2, 3, 32, 32)
    close(O.ddim_sample(sd, W.FR_UNET, s, 4, xT, cond=c), g["sample_S4"], 1e-4, 1e-4)
    ts = O.make_ddim_timesteps(200)

    def run3(eta, scale, noise):
        tab = O.make_ddim_tables(s["alphas_cumprod"], ts, eta)
        img = xT
        for i, step in enumerate(np.flip(ts)[:3]):
            idx = 200 - i - 1
            t = torch.full((2,), int(step), dtype=torch.long)
            if scale == 1.0:
                e = O.apply_model(sd, W.FR_UNET, img, t, [c])
            else:
                eu, ec = O.apply_model(sd, W.FR_UNET, torch.cat([img] * 2), torch.cat([t] * 2),
                                       [torch.cat([uc, c])]).chunk(2)
                e = O.cfg_combine(eu, ec, scale)
            img, _ = O.ddim_update(img, e, tab["a_t"][idx], tab["a_prev"][idx], tab["sigma_t"][idx],
                                   tab["sqrt_one_minus_at"][idx], None if noise is None else noise[i])
        return img

    close(run3(0.0, 1.0, None), g["s200_e0_cfg1"], 1e-4, 1e-4)
    close(run3(0.0, 3.0, None), g["s200_e0_cfg3"], 1e-4, 1e-4)
    close(run3(1.0, 1.0, T(g["s200_e1_noise"])), g["s200_e1_cfg1"], 1e-4, 1e-4)
    out = O.p_sample_loop(sd, W.FR_UNET, s, xT, cond=c, timesteps=3, noise=list(T(g["p_sample_loop_noise"])))
    close(out, g["p_sample_loop_T3"], 1e-4, 1e-4)


def test_g6_vqgan():
    g = golden("g6_vqgan.npz")
    sd = recipe(W.vqmodel_param_shapes(W.VQ_F4))
    z = rnd(61, 1, 3, 32, 32)
    zq, idx = O.vq_quantize(z, sd["quantize.embedding.weight"])
    assert np.array_equal(idx.numpy().astype(np.int32), g["vq_idx"].reshape(-1))
    close(zq, g["vq_zq"], 0, 1e-7)
    keys = {}
    W._vq_attn(keys, "", 512)
    close(O.vq_attn_block(recipe(keys, seed=1), "", rnd(62, 1, 512, 8, 8)), g["attn_block"], 1e-5, 2e-5)
    keys = {}
    W._vq_resnet(keys, "", 256, 128)
    close(O.vq_resnet_block(recipe(keys, seed=2), "", rnd(63, 1, 256, 8, 8)), g["resnet_block"], 1e-5, 2e-5)
    dec, _ = O.decode_first_stage(sd, W.VQ_F4, z)
    close(dec, g["decoded"].astype(np.float32), 2e-3, 2e-3)       # stored as fp16
    st = g["decoded_stats"]
    assert abs(dec.abs().max().item() - st[0]) < 1e-3 and abs(dec.std().item() - st[2]) < 1e-4
    img = torch.tanh(rnd(64, 1, 3, 128, 128))
    close(O.encode_first_stage(sd, W.VQ_F4, img), g["encoded"], 1e-4, 2e-4)


def test_g11_northstar_shapes():
    """The oracle against the real reference at BASELINE.json's metric shape: dim-4 / 16384-code quantiser (indices
    bit-exact), decode 4x64x64 -> 3x256x256, DDIMSampler.sample S=4 at 64x64x4."""
    g = golden("g11_northstar.npz")
    vsd = recipe(W.vqmodel_param_shapes(W.VQ_F4_256))
    z = rnd(112, 1, 4, 64, 64)
    zq, idx = O.vq_quantize(z, vsd["quantize.embedding.weight"])
    assert np.array_equal(idx.reshape(-1).numpy(), g["vq4_idx"].reshape(-1))
    close(zq, g["vq4_zq"], 0, 0)
    img, _ = O.decode_first_stage(vsd, W.VQ_F4_256, z)
    close(img, g["decoded256"], 1e-4, 2e-4)
    vsd128 = recipe(W.vqmodel_param_shapes(W.VQ_F4))
    img128, _ = O.decode_first_stage(vsd128, W.VQ_F4, rnd(61, 1, 3, 32, 32))
    close(img128, g["decoded128"], 1e-4, 2e-4)
    usd = recipe(W.unet_param_shapes(W.NS_UNET), gain=0.25)
    emb = T(W.synth_tensor("embedding.weight", (8, 512)))
    out = O.ddim_sample(usd, W.NS_UNET, O.register_schedule(**W.SCHEDULE), 4, rnd(111, 2, 4, 64, 64),
                        cond=emb[[3, 4]][:, None])
    close(out, g["sample_S4"], 1e-4, 1e-4)


def test_g7_talking_face():
    g = golden("g7_talking_face.npz")
    usd = recipe(W.unet_param_shapes(W.TF_UNET))
    out = O.apply_model(usd, W.TF_UNET, rnd(71, 2, 3, 32, 32), torch.tensor([11, 756]),
                        [rnd(72, 2, 1, 1024)], [rnd(73, 2, 6, 32, 32)])
    close(out, g["tf_eps"], 1e-4, 1e-4)
    asd = recipe(W.audio_attention_param_shapes(3))
    close(O.audio_temporal_attention(asd, rnd(74, 2, 3, 768)), g["audio_att"])
    vsd = recipe(W.vqmodel_param_shapes(W.VQ_F4))
    usd = recipe(W.unet_param_shapes(W.TF_UNET), gain=0.25)
    s = O.register_schedule(**W.SCHEDULE)
    Tn, S = 3, 4
    audio = rnd(75, Tn, 768)
    masked = torch.tanh(rnd(76, Tn, 3, 128, 128))
    masked[:, :, 70:, :] = -1.0
    ident = torch.tanh(rnd(77, 1, 3, 128, 128))
    c1 = T(W.synth_tensor("embedding.weight", (9, 256))[[4]][:, None])
    xid = O.encode_first_stage(vsd, W.VQ_F4, ident)
    close(xid, g["xid"], 1e-4, 2e-4)
    xT = rnd(78, Tn, 1, 3, 32, 32)
    for fixed, tag in ((False, "autoreg"), (True, "fixed")):
        fr = torch.cat(O.progressive_sampling(usd, W.TF_UNET, s, vsd, W.VQ_F4, asd, c1, xid, masked, audio, S, 1,
                                              xT, fixed_identity=fixed))
        close(fr, g[f"frames_{tag}"], 1e-4, 3e-4)
    assert not np.allclose(g["frames_autoreg"][1:], g["frames_fixed"][1:])
    close(g["frames_fixed_batched"], g["frames_fixed"], 1e-4, 3e-4)


def test_g8_ddim_inversion():
    g = golden("g8_inversion.npz")
    sd = recipe(W.unet_param_shapes(W.FR_UNET), gain=0.25)
    s = O.register_schedule(**W.SCHEDULE)
    c, uc = _fr_cond()
    assert np.array_equal(O.make_ddim_timesteps_strength(4, 1000, 0.5), g["timesteps"])
    x0 = rnd(81, 2, 3, 32, 32)
    for tag, scale in (("cfg1", 1.0), ("cfg3", 3.0)):
        img, lat = O.ddim_invert_and_regenerate(sd, W.FR_UNET, s, 4, x0, c, strength=0.5, scale=scale,
                                                uncond=uc if scale != 1.0 else None)
        close(lat, g[f"xlat_{tag}"], 1e-4, 1e-4)
        close(img, g[f"img_{tag}"], 1e-4, 1e-4)


def test_g9_p_losses_backward_pins_the_oracle_autograd():
    """Training row N1: loss and gradients of the reference's own LatentDiffusion.p_losses (ddpm.py:1014-1047,
    autograd through the reference UNet incl. its checkpointed transformer blocks) against autograd through the
    oracle's functional UNet -- the oracle's backward is what the HIP gradients are then checked against elementwise."""
    import torch.nn.functional as F
    g = golden("g9_p_losses.npz")
    cfg = dict(W.FR_UNET, model_channels=64, channel_mult=[1, 2], num_res_blocks=1, attention_resolutions=[2, 1])
    sd = W.synth_state_dict(W.unet_param_shapes(cfg))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x0, noise = rnd(101, 2, 3, 16, 16), rnd(102, 2, 3, 16, 16)
    c = rnd(103, 2, 1, 512).requires_grad_(True)
    t = torch.tensor([17, 803])
    sched = O.register_schedule(**W.SCHEDULE)
    a = sched["sqrt_alphas_cumprod"][t].view(-1, 1, 1, 1)
    b = sched["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    with torch.enable_grad():
        eps = O.unet_forward(sdg, cfg, a * x0 + b * noise, t, c)
        loss = F.mse_loss(eps, noise)
        loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * float(g["loss"])
    assert float(g["loss"]) == float(g["loss_simple"])
    torch.testing.assert_close(c.grad, torch.from_numpy(g["dcontext"]), rtol=2e-4, atol=1e-7)
    stats = dict(zip([str(n) for n in g["names"]], g["stats"]))
    assert set(stats) == set(sd)
    for k, (s_ref, n_ref) in stats.items():
        gr = sdg[k].grad
        if gr is None:                       # dead parameters of the single-token cross-attention
            assert n_ref == 0.0, k
            continue
        assert abs(gr.double().norm().item() - n_ref) <= 2e-4 * n_ref + 1e-12, (k, gr.double().norm().item(), n_ref)
        assert abs(gr.double().sum().item() - s_ref) <= 2e-4 * n_ref * gr.numel() ** 0.5 + 1e-12, k
    for k in g.files:
        if k.startswith("grad:"):
            torch.testing.assert_close(sdg[k[5:]].grad, torch.from_numpy(g[k]), rtol=2e-4, atol=1e-6 * float(np.abs(g[k]).max()))


def _oracle_diffclip(dtype=torch.float32):
    """The oracle's restatement of G10: 3 differentiable DDIM steps (ddim2.py:252-290, guidance 2 by batch doubling),
    differentiable decode (straight-through VQ), l2 image loss; returns everything autograd produced."""
    import torch.nn.functional as F
    from dsml_thesis_amd.schedule import ddim_step_table, make_ddim_timesteps_strength
    cfg = dict(W.FR_UNET, model_channels=64, channel_mult=[1, 2], num_res_blocks=1, attention_resolutions=[2, 1])
    sd = W.synth_state_dict(W.unet_param_shapes(cfg))
    vsd = W.synth_state_dict(W.vqmodel_param_shapes(W.VQ_F4))
    sdg = {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}
    v = {k: t.to(dtype) for k, t in vsd.items()}
    sched = O.register_schedule(**W.SCHEDULE)
    ts = make_ddim_timesteps_strength(3, 1000, 0.3)
    table = ddim_step_table(sched["alphas_cumprod"], ts, 0.0)
    x = rnd(401, 1, 3, 16, 16).to(dtype).requires_grad_(True)
    x0 = torch.tanh(rnd(402, 1, 3, 64, 64)).to(dtype)
    c, uc = rnd(403, 1, 1, 512).to(dtype), rnd(404, 1, 1, 512).to(dtype)
    with torch.enable_grad():
        xi = x
        for i in reversed(range(len(ts))):
            a_t, a_prev, _, s1m = (float(q) for q in table[i])
            tt = torch.full((1,), int(ts[i]))
            e2 = O.unet_forward(sdg, cfg, torch.cat([xi, xi]), torch.cat([tt, tt]), torch.cat([uc, c]))
            e_t = e2[:1] + 2.0 * (e2[1:] - e2[:1])
            pred_x0 = (xi - s1m * e_t) / a_t ** 0.5
            xi = a_prev ** 0.5 * pred_x0 + (1.0 - a_prev) ** 0.5 * e_t
        zq, _ = O.vq_quantize(xi.detach().float(), vsd["quantize.embedding.weight"])
        zst = xi + (zq.to(dtype) - xi).detach()
        img = O.decoder_forward(v, W.VQ_F4["ddconfig"], F.conv2d(zst, v["post_quant_conv.weight"], v["post_quant_conv.bias"]))
        loss = F.mse_loss(img, x0)
        loss.backward()
    return ts, xi.detach(), img.detach(), loss.detach(), x.grad, {k: t.grad for k, t in sdg.items()}


def test_g10_differentiable_ddim_pins_the_oracle():
    g = golden("g10_diffclip.npz")
    ts, z, img, loss, dx, grads = _oracle_diffclip()
    assert list(ts) == list(g["timesteps"])
    torch.testing.assert_close(z, torch.from_numpy(g["z"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(img, torch.from_numpy(g["image"]).float(), rtol=2e-3, atol=2e-3)      # stored as fp16
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    torch.testing.assert_close(dx, torch.from_numpy(g["dx"]), rtol=1e-3, atol=1e-6 * float(np.abs(g["dx"]).max()) + 1e-9)
    stats = dict(zip([str(n) for n in g["names"]], g["stats"]))
    for k, (_, n_ref) in stats.items():
        if grads[k] is None:
            assert n_ref == 0.0, k
        else:
            assert abs(grads[k].double().norm().item() - n_ref) <= 1e-3 * n_ref + 1e-12, k


def _options_inputs():
    usd = recipe(W.unet_param_shapes(W.FR_UNET), gain=0.25)
    emb = T(W.synth_tensor("embedding.weight", (8, 512)))
    c = emb[torch.tensor([1, 6])][:, None]
    uc = T(W.synth_tensor("uncond_embedding.weight", (1, 512)))[None].expand(2, 1, 512)
    code = T(W.synth_tensor("quantize.embedding.weight", (16384, 3)))
    xT, x0 = rnd(51, 2, 3, 32, 32), 0.5 * rnd(52, 2, 3, 32, 32)
    mask = (rnd(53, 2, 1, 32, 32) > 0).float()
    return usd, c, uc, code, xT, x0, mask


def test_g12_sampler_options():
    """The sampler options no shipped script sets (ddim.py:143-146,179-181,195-201; ddpm.py:1069-1072,1205-1208), against
    runs of the reference's own DDIMSampler.sample / p_sample_loop with those options (tools/make_golden.py --tree options)."""
    from tools.make_golden import ShiftCorrector
    g = golden("g12_sampler_options.npz")
    sched = O.register_schedule(**W.SCHEDULE)
    usd, c, uc, code, xT, x0, mask = _options_inputs()
    mine = O.ddim_sample(usd, W.FR_UNET, sched, 4, xT, cond=c, eta=1.0, scale=3.0, uncond=uc, noise=T(g["step_noise"]),
                         mask=mask, x0=x0, mask_noise=T(g["mask_noise"]), temperature=0.7)
    close(mine, g["ddim_mask_temp_cfg"], 1e-4, 1e-4)
    close(O.ddim_sample(usd, W.FR_UNET, sched, 4, xT, cond=c, quantize_codebook=code), g["ddim_quantize"], 1e-4, 1e-4)
    corr = ShiftCorrector()
    mine = O.ddim_sample(usd, W.FR_UNET, sched, 4, xT, cond=c, scale=3.0, uncond=uc,
                         score_fn=lambda e, x, t: corr.modify_score(None, e, x, t, None, strength=0.2))
    close(mine, g["ddim_corrector_cfg"], 1e-4, 1e-4)
    mine = O.p_sample_loop(usd, W.FR_UNET, sched, xT, cond=c, timesteps=3, noise=T(g["ddpm_noise"]), clip_denoised=True,
                           quantize_codebook=code, mask=mask, x0=x0, mask_noise=T(g["ddpm_mask_noise"]))
    close(mine, g["ddpm_clip_quant_mask"], 1e-4, 1e-4)


def test_g13_unconditional_unet_and_attention_block():
    """BASELINE configs[0] as worded (tests/golden/g13_config0.npz, from the real reference's unconditional LatentDiffusion):
    the oracle's AttentionBlock / QKVAttentionLegacy restatement (openaimodel.py:278-324,347-372) and the unconditional UNet
    built on it reproduce the reference's outputs.  (The 50-step chains of that file were checked against the oracle when the
    fixture was generated -- ~50 CPU evaluations at 64x64 each -- and are what the GPU tests are held to.)"""
    g = golden("g13_config0.npz")
    keys = {}
    W._attention_block(keys, "", 160)
    close(O.attention_block(recipe(keys, seed=11), "", rnd(131, 2, 160, 8, 8), 5), g["attention_block"], 1e-5, 2e-5)
    sd = recipe(W.unet_param_shapes(W.UNCOND_UNET), gain=0.25)
    out = O.unet_forward(sd, W.UNCOND_UNET, rnd(130, 2, 4, 64, 64), torch.tensor([7, 640]), None)
    close(out, g["uncond_eps"], 1e-4, 1e-4)
    assert g["uncond_ddim50"].shape == (1, 4, 64, 64) and g["ns_ddim50"].shape == (1, 4, 64, 64)


def test_g14_head_widths_and_original_steps():
    """g14 (tools/make_golden.py --tree variants, the real reference): UNets whose attention heads are 40 / 80 / 64 wide (num_heads = 4,
    num_head_channels = 64), with a one-token and a three-token context; `use_original_steps` updates of the talking-face sampler
    (ddim2cond.py:158-195) at eta 0 / 1 on four of the model's own timesteps."""
    g = golden("g14_variants.npz")
    x, t, ctx, ctx3 = rnd(150, 2, 3, 16, 16), torch.tensor([11, 870]), rnd(151, 2, 1, 512), rnd(152, 2, 3, 512)
    for tag, cfg in (("h40", W.H40_UNET), ("h64", W.H64_UNET)):
        sd = recipe(W.unet_param_shapes(cfg))
        lay = W.unet_layout(cfg)
        widths = sorted({l[3] for blk in lay["input"] + [lay["middle"]] + lay["output"] for l in blk if l[0] == "st"})
        assert widths == ([40, 80] if tag == "h40" else [64]), widths
        close(O.unet_forward(sd, cfg, x, t, ctx), g[tag + "_eps"], 2e-5, 2e-5)
        close(O.unet_forward(sd, cfg, x, t, ctx3), g[tag + "_eps_L3"], 2e-5, 2e-5)
    sched = O.register_schedule(**W.SCHEDULE)
    xs, eps = rnd(160, 2, 3, 32, 32), rnd(161, 2, 3, 32, 32)
    for eta in (0.0, 1.0):
        tabs = O.ddim_original_tables(sched, eta)
        for index in (0, 1, 437, 999):
            xp, p0 = O.p_sample_ddim_original(xs, eps, index, tabs, T(g[f"orig_eta{eta:g}_i{index}_noise"]))
            assert np.array_equal(xp.numpy(), g[f"orig_eta{eta:g}_i{index}_x_prev"]) and np.array_equal(p0.numpy(), g[f"orig_eta{eta:g}_i{index}_pred_x0"])


def test_g14_class_conditional_unet_with_scale_shift_norm_and_new_attention_order():
    """g14 `adm_eps`: the real UNetModel with use_scale_shift_norm, num_classes = 10 (label embedding added to the timestep
    embedding: the 'adm' conditioning key) and use_new_attention_order (QKVAttention) -- openaimodel.py:267-271,513-514,726-728,379-407."""
    g = golden("g14_variants.npz")
    sd = recipe(W.unet_param_shapes(W.ADM_UNET))
    x, t, y = rnd(153, 2, 3, 16, 16), torch.tensor([3, 512]), torch.tensor([7, 2])
    close(O.unet_forward(sd, W.ADM_UNET, x, t, None, y=y), g["adm_eps"], 2e-5, 2e-5)
    with pytest.raises(AssertionError):
        O.unet_forward(sd, W.ADM_UNET, x, t, None)


def test_g15_resblock_updown():
    """g15 (tools/make_golden.py --tree updown, the real reference): `resblock_updown=True` -- ResBlock(down=True) / ResBlock(up=True)
    where Downsample / Upsample would stand (openaimodel.py:570-584,660-674; _forward :256-261: avg_pool2d(2, 2) / nearest x2 on
    SiLU(GroupNorm(x)) and on the skip path) -- on the shipped spatial-transformer UNet (one ResBlock per level, 32x32) and, with
    use_scale_shift_norm, on the class-conditional UNet."""
    g = golden("g15_updown.npz")
    lay = W.unet_layout(W.UPDOWN_UNET)
    kinds = [l[0] for blk in lay["input"] + lay["output"] for l in blk]
    assert kinds.count("res_down") == 2 and kinds.count("res_up") == 2 and "down" not in kinds and "up" not in kinds
    sd = recipe(W.unet_param_shapes(W.UPDOWN_UNET))
    assert not any(k.endswith("op.weight") or k.endswith(".conv.weight") for k in sd)       # (no Downsample / Upsample convolutions)
    t, ctx = torch.tensor([5, 640]), rnd(171, 2, 1, 512)
    close(O.unet_forward(sd, W.UPDOWN_UNET, rnd(170, 2, 3, 32, 32), t, ctx), g["ud_eps"], 2e-5, 2e-5)
    sd = recipe(W.unet_param_shapes(W.UPDOWN_ADM_UNET))
    x, t, y = rnd(173, 2, 3, 16, 16), torch.tensor([3, 512]), torch.tensor([7, 2])
    close(O.unet_forward(sd, W.UPDOWN_ADM_UNET, x, t, None, y=y), g["ud_adm_eps"], 2e-5, 2e-5)


def test_g15_class_conditional_model_under_the_samplers():
    """g15 `adm_*`: the real LatentDiffusion(conditioning_key='adm') under the real DDIMSampler.sample (plain / CFG 3) and
    p_sample_loop -- the labels travel as c_crossattn = [y] and reach the UNet as y (ddpm.py:893-994,1417-1419)."""
    g = golden("g15_updown.npz")
    cfg = W.UPDOWN_ADM_UNET
    sd = recipe(W.unet_param_shapes(cfg), gain=0.25)
    sched = O.register_schedule(**W.SCHEDULE)
    xT, y, uy = rnd(176, 2, 3, 16, 16), torch.tensor([7, 2]), torch.tensor([0, 0])
    close(O.ddim_sample(sd, cfg, sched, 4, xT, cond=y), g["adm_ddim4"], 1e-4, 1e-4)
    close(O.ddim_sample(sd, cfg, sched, 4, xT, cond=y, scale=3.0, uncond=uy), g["adm_ddim4_cfg3"], 1e-4, 1e-4)
    close(O.p_sample_loop(sd, cfg, sched, xT, cond=y, timesteps=3, noise=list(T(g["adm_ddpm3_noise"]))), g["adm_ddpm3"], 1e-4, 1e-4)


def test_g16_stochastic_encode_and_decode():
    """g16 (tools/make_golden.py --tree sdedit): the real talking-face DDIMSampler.stochastic_encode / decode (ddim2cond.py:198-250)
    around the real LatentDiffusion -- encode bit for bit (the sampler's own float32-root tables), the three-step decodes at 1e-4."""
    g = golden("g16_sdedit.npz")
    sched = O.register_schedule(**W.SCHEDULE)
    x0, nz = rnd(180, 2, 3, 32, 32), rnd(181, 2, 3, 32, 32)
    c12, c34 = rnd(182, 2, 1, 1024), rnd(183, 2, 6, 32, 32)
    assert np.array_equal(O.stochastic_encode(sched, 5, x0, torch.tensor([3, 1]), nz).numpy(), g["enc"])
    assert np.array_equal(O.stochastic_encode(sched, 5, x0, torch.tensor([640, 7]), nz, use_original_steps=True).numpy(), g["enc_orig"])
    sd = recipe(W.unet_param_shapes(W.TF_UNET), gain=0.25)
    x_lat = O.stochastic_encode(sched, 5, x0, torch.tensor([2, 2]), nz)
    close(O.ddim_decode(sd, W.TF_UNET, sched, 5, x_lat, 3, cond=c12, c_concat=c34), g["dec3"], 1e-4, 1e-4)
    close(O.ddim_decode(sd, W.TF_UNET, sched, 5, x_lat, 3, cond=c12, c_concat=c34, eta=1.0, noise=list(T(g["dec3_eta1_noise"]))),
          g["dec3_eta1"], 1e-4, 1e-4)


def test_g17_latent_manipulation():
    """g17 (tools/make_golden.py --tree manip): DDIMSampler.latent_manipulation of the driver script latent_manipulation.py
    (:420-490) on the real LatentDiffusion -- inversion under the source label, regeneration under the target label."""
    g = golden("g17_manipulation.npz")
    sched = O.register_schedule(**W.SCHEDULE)
    sd = recipe(W.unet_param_shapes(W.FR_UNET), gain=0.25)
    x0, c_src, c_trg, uc = rnd(190, 2, 3, 32, 32), T(g["c_src"]), T(g["c_trg"]), T(g["uc"])
    img, lat = O.ddim_invert_and_regenerate(sd, W.FR_UNET, sched, 4, x0, c_src, strength=0.5, cond_trg=c_trg)
    close(lat, g["xlat_cfg1"], 1e-4, 1e-4)
    close(img, g["img_cfg1"], 1e-4, 1e-4)
    img3, lat3 = O.ddim_invert_and_regenerate(sd, W.FR_UNET, sched, 4, x0, c_src, strength=0.5, scale=3.0, uncond=uc, cond_trg=c_trg)
    close(lat3, g["xlat_cfg3"], 1e-4, 1e-4)
    close(img3, g["img_cfg3"], 1e-4, 1e-4)
