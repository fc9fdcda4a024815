"""TEST INFRASTRUCTURE ONLY (oracle/) -- CPU restatement of the reference sampling path.

Pure PyTorch-CPU fp32, functional style over a flat state-dict keyed with the
reference's parameter names.  This is the *checker*: only tests/, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; the product package never does.

Parity status: PINNED.  `tools/make_golden.py` imports the real reference modules from
/root/reference (this container only), checks every function below against them on the
same seeded inputs, and commits the reference's outputs under tests/golden/*.npz; the
CPU test-suite re-checks this file against those fixtures (tests/test_oracle_golden.py).
The reference itself ships no tests or golden vectors.

Each function cites the reference code it restates (paths relative to
/root/reference/face_reenactment unless prefixed TF = /root/reference/talking_face).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .weights import unet_layout


# ============================================================================ schedules
def make_beta_schedule(timesteps=1000, linear_start=1e-4, linear_end=2e-2):
    """'linear' schedule, ldm/modules/diffusionmodules/util.py:21-25 (float64)."""
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2


def register_schedule(timesteps=1000, linear_start=1e-4, linear_end=2e-2, v_posterior=0.0):
    """DDPM.register_schedule, ldm/models/diffusion/ddpm.py:117-169.  float64 math, float32 buffers."""
    betas = make_beta_schedule(timesteps, linear_start, linear_end)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    post_var = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
    return dict(
        betas=f32(betas), alphas_cumprod=f32(ac), alphas_cumprod_prev=f32(ac_prev),
        sqrt_alphas_cumprod=f32(np.sqrt(ac)), sqrt_one_minus_alphas_cumprod=f32(np.sqrt(1.0 - ac)),
        log_one_minus_alphas_cumprod=f32(np.log(1.0 - ac)),
        sqrt_recip_alphas_cumprod=f32(np.sqrt(1.0 / ac)),
        sqrt_recipm1_alphas_cumprod=f32(np.sqrt(1.0 / ac - 1)),
        posterior_variance=f32(post_var),
        posterior_log_variance_clipped=f32(np.log(np.maximum(post_var, 1e-20))),
        posterior_mean_coef1=f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        posterior_mean_coef2=f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    )


def make_ddim_timesteps(num_ddim, num_ddpm=1000):
    """'uniform' discretisation, util.py:46-60."""
    c = num_ddpm // num_ddim
    return np.asarray(list(range(0, num_ddpm, c))) + 1


def make_ddim_tables(alphas_cumprod_f32, ddim_timesteps, eta):
    """util.py:63-74 + DDIMSampler.make_schedule ddim.py:42-50.

    `alphas_cumprod_f32` is the float32 buffer; the reference indexes the *torch float32*
    tensor with a numpy index array (-> float32 `alphas`), builds `alphas_prev` through
    python floats (float64 ndarray) and hence `sigmas` in float64.
    Returns the four per-index tables squeezed to float32 exactly like `torch.full` does
    (ddim.py:188-191).
    """
    ac = torch.as_tensor(alphas_cumprod_f32, dtype=torch.float32)
    alphas = ac[ddim_timesteps]                                            # f32 tensor
    alphas_prev = np.asarray([ac[0].item()] + ac[ddim_timesteps[:-1]].tolist())  # f64 ndarray
    # mixed precision exactly as the reference's torch/numpy dispatch does it (util.py:69):
    # `ndarray / Tensor` resolves to Tensor.__rtruediv__ = reciprocal(float32) * ndarray(float64);
    # `Tensor / ndarray` promotes to float64 before dividing.
    a64 = alphas.double().numpy()
    recip_one_minus_a = (1 - alphas).reciprocal().double().numpy()
    sigmas = eta * np.sqrt((1 - alphas_prev) * recip_one_minus_a * (1 - a64 / alphas_prev))
    sqrt_1ma = np.sqrt(1.0 - alphas.numpy())                               # f32
    return dict(
        a_t=alphas.numpy().astype(np.float32),
        a_prev=alphas_prev.astype(np.float32),
        sigma_t=np.asarray(sigmas).astype(np.float32),
        sqrt_one_minus_at=sqrt_1ma.astype(np.float32),
    )


def ddim_update(x, e_t, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None):
    """DDIMSampler.p_sample_ddim, ddim.py:193-203 (float32 scalars broadcast)."""
    a_t = torch.tensor(a_t, dtype=torch.float32)
    a_prev = torch.tensor(a_prev, dtype=torch.float32)
    sigma_t = torch.tensor(sigma_t, dtype=torch.float32)
    s1 = torch.tensor(sqrt_one_minus_at, dtype=torch.float32)
    pred_x0 = (x - s1 * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt
    if noise is not None:          # eta == 0 -> sigma_t == 0 and the reference adds an exact zero
        x_prev = x_prev + sigma_t * noise
    return x_prev, pred_x0


def ddim_original_tables(sched, eta):
    """`use_original_steps` tables: talking_face/ldm/models/diffusion/ddim2cond.py:175-178 read `model.alphas_cumprod`,
    `model.alphas_cumprod_prev`, `model.sqrt_one_minus_alphas_cumprod` and the sampler's `ddim_sigmas_for_original_num_steps`
    = eta sqrt((1 - a_prev) / (1 - a) (1 - a / a_prev)), formed in float32 from the float32 buffers (make_schedule,
    ddim2cond.py:31-53 / ddim.py:31-53).  (face_reenactment/.../ddim.py:186 reads the sigmas off `self.model`, which never defines
    them: that copy raises AttributeError for use_original_steps=True.)  -> dict of four float32 tensors [T]."""
    a, ap = sched["alphas_cumprod"].float(), sched["alphas_cumprod_prev"].float()
    sig = eta * torch.sqrt((1 - ap) / (1 - a) * (1 - a / ap))
    return dict(a_t=a, a_prev=ap, sigma_t=sig.float(), sqrt_one_minus_at=sched["sqrt_one_minus_alphas_cumprod"].float())


def p_sample_ddim_original(x, e_t, index, tabs, noise=None):
    """One `use_original_steps` update (ddim2cond.py:179-195) on the model's own timestep `index`."""
    return ddim_update(x, e_t, tabs["a_t"][index].item(), tabs["a_prev"][index].item(), tabs["sigma_t"][index].item(),
                       tabs["sqrt_one_minus_at"][index].item(), noise)


def cfg_combine(e_uncond, e_cond, scale):
    """ddim.py:177."""
    return e_uncond + scale * (e_cond - e_uncond)


def ddpm_update(sched, x, eps, t, noise):
    """LatentDiffusion.p_mean_variance + p_sample (eps-param, no clip): ddpm.py:215-228,1049-1109."""
    sh = (x.shape[0],) + (1,) * (x.dim() - 1)
    g = lambda name: sched[name].gather(-1, t).reshape(sh)
    x0 = g("sqrt_recip_alphas_cumprod") * x - g("sqrt_recipm1_alphas_cumprod") * eps
    mean = g("posterior_mean_coef1") * x0 + g("posterior_mean_coef2") * x
    logvar = g("posterior_log_variance_clipped")
    nonzero = (1 - (t == 0).float()).reshape(sh)
    return mean + nonzero * (0.5 * logvar).exp() * noise


# ============================================================================ UNet pieces
def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def gn_silu(x, w, b, eps=1e-5, silu=True, groups=32):
    """GroupNorm32 + SiLU: util.py:199-216, openaimodel.py:201-203."""
    xf = x if x.dtype == torch.float64 else x.float()   # GroupNorm32 computes in float32; float64 = gradient tests
    y = F.group_norm(xf, groups, w, b, eps)
    return F.silu(y) if silu else y


def resblock(sd, p, x, emb, scale_shift=False, updown=None):
    """ResBlock._forward: openaimodel.py:255-275; scale_shift = use_scale_shift_norm (:267-271); updown = "up" / "down": the
    ResBlock(up=True / down=True) of resblock_updown (:256-261) -- h_upd / x_upd are Upsample / Downsample WITHOUT a convolution
    (:207-216: nearest x2, :110-118, and avg_pool2d(2, 2), :143-160), applied to SiLU(GroupNorm(x)) before the first convolution
    and to x on the skip path."""
    h = gn_silu(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"])
    if updown == "up":
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    elif updown == "down":
        h, x = F.avg_pool2d(h, 2, 2), F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    if scale_shift:
        scale, shift = torch.chunk(e[:, :, None, None], 2, dim=1)
        h = F.group_norm(h.float(), 32, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"], 1e-5).type(h.dtype) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = h + e[:, :, None, None]
        h = gn_silu(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"])
    h = F.conv2d(h, sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if p + "skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + "skip_connection.weight"], sd[p + "skip_connection.bias"])
    return x + h


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward, attention.py:170-193 (mask=None)."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(ctx, sd[p + "to_k.weight"])
    v = F.linear(ctx, sd[p + "to_v.weight"])
    b, n, c = q.shape
    d = c // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def geglu_ff(sd, p, x):
    """FeedForward(glu=True): attention.py:37-64."""
    h = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    h = a * F.gelu(gate)
    return F.linear(h, sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def transformer_block(sd, p, x, context, heads):
    """BasicTransformerBlock._forward, attention.py:211-215."""
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-5)
    x = cross_attention(sd, p + "attn1.", ln(x, "norm1"), None, heads) + x
    x = cross_attention(sd, p + "attn2.", ln(x, "norm2"), context, heads) + x
    x = geglu_ff(sd, p + "ff.", ln(x, "norm3")) + x
    return x


def spatial_transformer(sd, p, x, context, heads, depth=1):
    """SpatialTransformer.forward, attention.py:250-261 (GroupNorm eps 1e-6, attention.py:76-77)."""
    b, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    x = F.conv2d(x, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    x = x.reshape(b, x.shape[1], h * w).permute(0, 2, 1)
    for d in range(depth):
        x = transformer_block(sd, f"{p}transformer_blocks.{d}.", x, context, heads)
    x = x.permute(0, 2, 1).reshape(b, -1, h, w)
    x = F.conv2d(x, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return x + x_in


def attention_block(sd, p, x, heads, new_order=False):
    """AttentionBlock._forward + QKVAttentionLegacy.forward, openaimodel.py:316-324,358-372: GroupNorm32 (eps 1e-5, no
    activation), qkv Conv1d, heads split BEFORE q/k/v ([head][q|k|v][ch] channel order), q and k each scaled by ch^-1/4,
    softmax over the keys, proj_out Conv1d, residual."""
    b, c, h, w = x.shape
    xr = x.reshape(b, c, h * w)
    qkv = F.conv1d(F.group_norm(xr.float(), 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5).type(xr.dtype),
                   sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    ch = c // heads
    if new_order:      # QKVAttention (use_new_attention_order, openaimodel.py:379-407): q | k | v first, heads inside each
        q, k, v = (t.reshape(b * heads, ch, h * w) for t in qkv.chunk(3, dim=1))
    else:
        q, k, v = qkv.reshape(b * heads, 3 * ch, h * w).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    weight = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale).float(), dim=-1).type(qkv.dtype)
    a = torch.einsum("bts,bcs->bct", weight, v).reshape(b, c, h * w)
    return (xr + F.conv1d(a, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])).reshape(b, c, h, w)


def _run_layers(sd, cfg, prefix, layers, h, emb, context):
    """TimestepEmbedSequential.forward, openaimodel.py:80-88."""
    for j, l in enumerate(layers):
        p = f"{prefix}{j}."
        if l[0] == "conv":
            h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
        elif l[0] == "res":
            h = resblock(sd, p, h, emb, bool(cfg.get("use_scale_shift_norm", False)))
        elif l[0] in ("res_down", "res_up"):      # resblock_updown, openaimodel.py:570-584,660-674
            h = resblock(sd, p, h, emb, bool(cfg.get("use_scale_shift_norm", False)), updown=l[0][4:])
        elif l[0] == "st":
            h = spatial_transformer(sd, p, h, context, l[2], cfg.get("transformer_depth", 1))
        elif l[0] == "attn":
            h = attention_block(sd, p, h, l[2], bool(cfg.get("use_new_attention_order", False)))
        elif l[0] == "down":   # Downsample, openaimodel.py:150-160
            h = F.conv2d(h, sd[p + "op.weight"], sd[p + "op.bias"], stride=2, padding=1)
        elif l[0] == "up":     # Upsample, openaimodel.py:107-118
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)
    return h


def unet_forward(sd, cfg, x, timesteps, context=None, y=None):
    """UNetModel.forward, openaimodel.py:710-742 (y: class labels of a num_classes UNet, :726-728)."""
    lay = unet_layout(cfg)
    wdt = sd["time_embed.0.weight"].dtype      # float32 (the reference's self.dtype); float64 only in gradient tests
    t_emb = timestep_embedding(timesteps, cfg["model_channels"]).to(wdt)
    emb = F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    assert (y is not None) == (cfg.get("num_classes") is not None), "must specify y if and only if the model is class-conditional"
    if y is not None:
        emb = emb + sd["label_emb.weight"][y]
    hs = []
    h = x.to(wdt)                              # h = x.type(self.dtype), openaimodel.py:728
    for i, layers in enumerate(lay["input"]):
        h = _run_layers(sd, cfg, f"input_blocks.{i}.", layers, h, emb, context)
        hs.append(h)
    h = _run_layers(sd, cfg, "middle_block.", lay["middle"], h, emb, context)
    for i, layers in enumerate(lay["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, cfg, f"output_blocks.{i}.", layers, h, emb, context)
    h = gn_silu(h, sd["out.0.weight"], sd["out.0.bias"])
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def apply_model(sd, cfg, x, t, c_crossattn=None, c_concat=None):
    """DiffusionWrapper.forward: ddpm.py:1404-1423 ('crossattn'); TF ddpm2cond.py:1307-1315
    (concat on channels *and* cross-attention)."""
    if cfg.get("num_classes") is not None:             # conditioning_key 'adm' (ddpm.py:1417-1419): the conditioning IS y
        return unet_forward(sd, cfg, x, t, None, y=c_crossattn[0])
    if c_concat is not None:
        x = torch.cat([x] + list(c_concat), dim=1)
    cc = None if c_crossattn is None else torch.cat(list(c_crossattn), 1)
    return unet_forward(sd, cfg, x, t, cc)


# ============================================================================ samplers
def ddim_sample(sd, cfg, sched, S, x_T, cond=None, c_concat=None, eta=0.0, scale=1.0, uncond=None,
                noise=None, return_all=False, mask=None, x0=None, mask_noise=None, temperature=1.0,
                quantize_codebook=None, score_fn=None):
    """DDIMSampler.sample/ddim_sampling/p_sample_ddim: ddim.py:56-203 (FR, CFG by batch doubling)
    and TF ddim2cond.py:56-195 (scale==1 path; c_concat = 'motion_&_id').

    `noise`: optional (S, *x.shape) pre-generated standard-normal noise used when eta>0
    (the reference draws torch.randn on the device, which no other backend can reproduce).
    Options of ddim.py:112-203 no shipped script sets: `mask`/`x0` (+ `mask_noise`, q_sample's per-step draw) blend
    `q_sample(x0, t) * mask + (1 - mask) * img` before each step (:143-146); `temperature` scales the step noise (:199);
    `quantize_codebook` snaps pred_x0 to the codebook before x_prev is formed (:195-196); `score_fn(e_t, x, t)` stands for
    score_corrector.modify_score (:179-181).
    """
    ts = make_ddim_timesteps(S, sched["betas"].shape[0])
    tab = make_ddim_tables(sched["alphas_cumprod"], ts, eta)
    img = x_T
    b = x_T.shape[0]
    traj = []
    for i, step in enumerate(np.flip(ts)):
        index = S - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        cc = None if c_concat is None else [c_concat]
        if mask is not None:
            sh = (b, 1, 1, 1)
            img_orig = (sched["sqrt_alphas_cumprod"].gather(-1, t).reshape(sh) * x0 +
                        sched["sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(sh) * mask_noise[i])
            img = img_orig * mask + (1. - mask) * img
        if uncond is None or scale == 1.0:
            e_t = apply_model(sd, cfg, img, t, None if cond is None else [cond], cc)
        else:
            x_in = torch.cat([img] * 2)
            t_in = torch.cat([t] * 2)
            c_in = torch.cat([uncond, cond])
            cc2 = None if c_concat is None else [torch.cat([c_concat] * 2)]
            e_u, e_c = apply_model(sd, cfg, x_in, t_in, [c_in], cc2).chunk(2)
            e_t = cfg_combine(e_u, e_c, scale)
        if score_fn is not None:
            e_t = score_fn(e_t, img, t)
        nz = None if noise is None else noise[i] * temperature
        x_in_step = img
        img, pred_x0 = ddim_update(img, e_t, tab["a_t"][index], tab["a_prev"][index],
                                   tab["sigma_t"][index], tab["sqrt_one_minus_at"][index], nz)
        if quantize_codebook is not None:                   # x_prev re-formed from the quantised x0 estimate
            pred_q, _ = vq_quantize(pred_x0, quantize_codebook)
            a_prev, sig = torch.as_tensor(tab["a_prev"][index]).float(), torch.as_tensor(tab["sigma_t"][index]).float()
            img = a_prev.sqrt() * pred_q + (1. - a_prev - sig ** 2).sqrt() * e_t + (0. if nz is None else sig * nz)
        if return_all:
            traj.append(img)
    return (img, traj) if return_all else img


def stochastic_encode(sched, S, x0, t, noise, use_original_steps=False):
    """DDIMSampler.stochastic_encode, ddim.py:205-219 / ddim2cond.py:198-212: q_sample on the DDIM subsequence's alphas (t indexes
    the S entries of the subsequence) or, use_original_steps, on the model's own schedule (eta does not enter)."""
    if use_original_steps:
        # the SAMPLER's buffers (make_schedule, ddim.py:35-36): float32 square roots of the float32 alphas_cumprod, one ulp
        # away from the model's own buffers in places (those are rounded from float64 roots)
        a, b = torch.sqrt(sched["alphas_cumprod"]), torch.sqrt(1. - sched["alphas_cumprod"])
    else:
        tab = make_ddim_tables(sched["alphas_cumprod"], make_ddim_timesteps(S, sched["betas"].shape[0]), 0.0)
        a, b = torch.sqrt(torch.as_tensor(tab["a_t"])), torch.as_tensor(tab["sqrt_one_minus_at"])
    sh = (x0.shape[0], 1, 1, 1)
    return a.gather(-1, t).reshape(sh) * x0 + b.gather(-1, t).reshape(sh) * noise


def ddim_decode(sd, cfg, sched, S, x_latent, t_start, cond=None, c_concat=None, eta=0.0, scale=1.0, uncond=None, noise=None):
    """DDIMSampler.decode, ddim2cond.py:230-250 (the second half of an img2img / SDEdit edit): the DDIM updates of the FIRST t_start
    entries of the subsequence, from index t_start - 1 down to 0, through p_sample_ddim."""
    ts = make_ddim_timesteps(S, sched["betas"].shape[0])
    tab = make_ddim_tables(sched["alphas_cumprod"], ts, eta)
    img, b = x_latent, x_latent.shape[0]
    for i, step in enumerate(np.flip(ts[:t_start])):
        index = t_start - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        cc = None if c_concat is None else [c_concat]
        if uncond is None or scale == 1.0:
            e_t = apply_model(sd, cfg, img, t, None if cond is None else [cond], cc)
        else:
            cc2 = None if c_concat is None else [torch.cat([c_concat] * 2)]
            e_u, e_c = apply_model(sd, cfg, torch.cat([img] * 2), torch.cat([t] * 2), [torch.cat([uncond, cond])], cc2).chunk(2)
            e_t = cfg_combine(e_u, e_c, scale)
        img, _ = ddim_update(img, e_t, tab["a_t"][index], tab["a_prev"][index], tab["sigma_t"][index], tab["sqrt_one_minus_at"][index],
                             None if noise is None else noise[i])
    return img


def make_ddim_timesteps_strength(num_ddim, num_ddpm=1000, strength=1.0):
    """compute_latents.py:52-73 (strength-scaled 'uniform' schedule of the latent-manipulation scripts)."""
    ts = np.linspace(0, 1, num_ddim) * int(num_ddpm * strength)
    return np.asarray([1] + [int(s) for s in list(ts)][1:])


def ddim_invert_and_regenerate(sd, cfg, sched, S, x0, cond, strength=0.5, scale=1.0, uncond=None, cond_trg=None):
    """DDIMSampler.compute_latents + q_sample_ddim, compute_latents.py:297-406: forward DDIM (inversion) over the
    strength-scaled timesteps, then the reverse loop.  Returns (img, x_latent).  cond_trg: DDIMSampler.latent_manipulation
    (latent_manipulation.py:420-490) -- the same two loops, inverted under `cond` (c_src), regenerated under `cond_trg`."""
    ts = make_ddim_timesteps_strength(S, sched["betas"].shape[0], strength)
    ac = torch.as_tensor(sched["alphas_cumprod"], dtype=torch.float32)
    alphas = ac[ts]                                                        # f32 tensor
    alphas_prev = np.asarray([ac[0].item()] + ac[ts[:-1]].tolist())        # f64 ndarray
    s1m = np.sqrt(1.0 - alphas.numpy())                                    # f32
    s1m_prev = np.sqrt(1.0 - alphas_prev)                                  # f64
    b = x0.shape[0]

    def eps_of(x, t, c=None):
        c = cond if c is None else c
        if uncond is None or scale == 1.0:
            return apply_model(sd, cfg, x, t, [c])
        e_u, e_c = apply_model(sd, cfg, torch.cat([x] * 2), torch.cat([t] * 2), [torch.cat([uncond, c])]).chunk(2)
        return cfg_combine(e_u, e_c, scale)

    f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
    x = x0.clone()
    for i, step in enumerate(ts):                                          # forward DDIM, :343-349
        t = torch.full((b,), int(step), dtype=torch.long)
        e = eps_of(x, t)
        at, at_next = f32(alphas_prev[i]), f32(alphas[i].item())
        pred_x0 = (x - f32(s1m_prev[i]) * e) / at.sqrt()
        x = at_next.sqrt() * pred_x0 + (1.0 - at_next).sqrt() * e
    x_lat = x.clone()
    img = x_lat
    for i, step in enumerate(np.flip(ts)):                                 # reverse DDIM, :351-360 (eta = 0)
        index = S - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        e = eps_of(img, t, cond_trg)
        img, _ = ddim_update(img, e, alphas[index].item(), np.float32(alphas_prev[index]), 0.0, s1m[index])
    return img, x_lat


def p_sample_loop(sd, cfg, sched, x_T, cond=None, timesteps=None, noise=None, clip_denoised=False,
                  quantize_codebook=None, mask=None, x0=None, mask_noise=None):
    """LatentDiffusion.p_sample_loop, ddpm.py:1167-1216 (clip_denoised False by default, ddpm.py:463).  Options:
    clamp / quantise the x0 estimate inside p_mean_variance (:1069-1072), inpainting blend after each step (:1205-1208)."""
    T = sched["betas"].shape[0] if timesteps is None else timesteps
    img = x_T
    b = x_T.shape[0]
    sh = (b, 1, 1, 1)
    g = lambda name, t: sched[name].gather(-1, t).reshape(sh)
    for k, i in enumerate(reversed(range(0, T))):
        t = torch.full((b,), i, dtype=torch.long)
        eps = apply_model(sd, cfg, img, t, None if cond is None else [cond])
        nz = torch.zeros_like(img) if noise is None else noise[k]
        if clip_denoised or quantize_codebook is not None:
            xr = g("sqrt_recip_alphas_cumprod", t) * img - g("sqrt_recipm1_alphas_cumprod", t) * eps
            if clip_denoised:
                xr = xr.clamp(-1., 1.)
            if quantize_codebook is not None:
                xr, _ = vq_quantize(xr, quantize_codebook)
            mean = g("posterior_mean_coef1", t) * xr + g("posterior_mean_coef2", t) * img
            img = mean + (1 - (t == 0).float()).reshape(sh) * (0.5 * g("posterior_log_variance_clipped", t)).exp() * nz
        else:
            img = ddpm_update(sched, img, eps, t, nz)
        if mask is not None:
            img_orig = g("sqrt_alphas_cumprod", t) * x0 + g("sqrt_one_minus_alphas_cumprod", t) * mask_noise[k]
            img = img_orig * mask + (1. - mask) * img
    return img


# ============================================================================ VQGAN first stage
def vq_quantize(z, codebook):
    """VectorQuantizer2.forward, taming/modules/vqvae/quantize.py:271-312 -> (z_q, indices)."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, codebook.shape[1])
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) \
        - 2 * torch.einsum("bd,dn->bn", zf, codebook.t())
    idx = torch.argmin(d, dim=1)
    z_q = codebook[idx].view(zp.shape)
    z_q = zp + (z_q - zp)
    return z_q.permute(0, 3, 1, 2).contiguous(), idx


def _vq_gn(sd, p, x, silu):
    y = F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)  # model.py:38-39
    return y * torch.sigmoid(y) if silu else y


def vq_resnet_block(sd, p, x):
    """ResnetBlock.forward (temb None), model.py:118-141."""
    h = _vq_gn(sd, p + "norm1", x, True)
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = _vq_gn(sd, p + "norm2", h, True)
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"])
    return x + h


def vq_attn_block(sd, p, x):
    """AttnBlock.forward, model.py:178-202 (single head, logits scaled by C^-0.5)."""
    h_ = _vq_gn(sd, p + "norm", x, False)
    q = F.conv2d(h_, sd[p + "q.weight"], sd[p + "q.bias"])
    k = F.conv2d(h_, sd[p + "k.weight"], sd[p + "k.bias"])
    v = F.conv2d(h_, sd[p + "v.weight"], sd[p + "v.bias"])
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    h_ = F.conv2d(h_, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return x + h_


def decoder_forward(sd, dd, z, prefix="decoder."):
    """Decoder.forward, model.py:535-568."""
    nres = len(dd["ch_mult"])
    curr = dd["resolution"] // 2 ** (nres - 1)
    h = F.conv2d(z, sd[prefix + "conv_in.weight"], sd[prefix + "conv_in.bias"], padding=1)
    h = vq_resnet_block(sd, prefix + "mid.block_1.", h)
    h = vq_attn_block(sd, prefix + "mid.attn_1.", h)
    h = vq_resnet_block(sd, prefix + "mid.block_2.", h)
    for lvl in reversed(range(nres)):
        for ib in range(dd["num_res_blocks"] + 1):
            h = vq_resnet_block(sd, f"{prefix}up.{lvl}.block.{ib}.", h)
            if curr in dd["attn_resolutions"]:
                h = vq_attn_block(sd, f"{prefix}up.{lvl}.attn.{ib}.", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")   # model.py:54
            h = F.conv2d(h, sd[f"{prefix}up.{lvl}.upsample.conv.weight"],
                         sd[f"{prefix}up.{lvl}.upsample.conv.bias"], padding=1)
            curr *= 2
    h = _vq_gn(sd, prefix + "norm_out", h, True)
    return F.conv2d(h, sd[prefix + "conv_out.weight"], sd[prefix + "conv_out.bias"], padding=1)


def encoder_forward(sd, dd, x, prefix="encoder."):
    """Encoder.forward, model.py:434-459 (asymmetric-pad stride-2 downsample, model.py:72-76)."""
    nres = len(dd["ch_mult"])
    curr = dd["resolution"]
    h = F.conv2d(x, sd[prefix + "conv_in.weight"], sd[prefix + "conv_in.bias"], padding=1)
    for lvl in range(nres):
        for ib in range(dd["num_res_blocks"]):
            h = vq_resnet_block(sd, f"{prefix}down.{lvl}.block.{ib}.", h)
            if curr in dd["attn_resolutions"]:
                h = vq_attn_block(sd, f"{prefix}down.{lvl}.attn.{ib}.", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"{prefix}down.{lvl}.downsample.conv.weight"],
                         sd[f"{prefix}down.{lvl}.downsample.conv.bias"], stride=2, padding=0)
            curr //= 2
    h = vq_resnet_block(sd, prefix + "mid.block_1.", h)
    h = vq_attn_block(sd, prefix + "mid.attn_1.", h)
    h = vq_resnet_block(sd, prefix + "mid.block_2.", h)
    h = _vq_gn(sd, prefix + "norm_out", h, True)
    return F.conv2d(h, sd[prefix + "conv_out.weight"], sd[prefix + "conv_out.bias"], padding=1)


def decode_first_stage(sd, fs, z, scale_factor=1.0, force_not_quantize=False):
    """LatentDiffusion.decode_first_stage -> VQModelInterface.decode:
    ddpm.py:706-764, autoencoder.py:274-282."""
    z = 1.0 / scale_factor * z
    idx = None
    if not force_not_quantize:
        z, idx = vq_quantize(z, sd["quantize.embedding.weight"])
    q = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return decoder_forward(sd, fs["ddconfig"], q), idx


def encode_first_stage(sd, fs, x):
    """VQModelInterface.encode, autoencoder.py:269-272 (no quantisation)."""
    h = encoder_forward(sd, fs["ddconfig"], x)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def postprocess_frames(x):
    """sample_affectnet.py:127,132: clamp((x+1)/2,0,1), NCHW->NHWC."""
    return torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0).permute(0, 2, 3, 1).contiguous()


# ============================================================================ TF conditioning
def audio_temporal_attention(sd, x, prefix=""):
    """Conv1DTemporalAttention.forward, TF ldm/modules/encoders/modules.py:103-113.
    x: (b, T, 768) -> (b, 1, 768)."""
    b, T, _ = x.shape
    xt = x.transpose(1, 2)
    h = xt
    for i in range(5):
        h = F.conv1d(h, sd[f"{prefix}attentionConvNet.{2 * i}.weight"],
                     sd[f"{prefix}attentionConvNet.{2 * i}.bias"], padding=1)
        h = F.leaky_relu(h, 0.02)
    a = F.linear(h.view(b, T), sd[prefix + "attentionNet.0.weight"], sd[prefix + "attentionNet.0.bias"])
    a = F.softmax(a, dim=1).view(b, T, 1)
    return torch.bmm(xt, a).view(b, -1).unsqueeze(1)


def audio_window_indices(frame_idx, num_frames, audio_window):
    """TF progressive_sampling_difftalk.py:287."""
    return [min(max(frame_idx + i, 0), num_frames - 1) for i in range(-audio_window, audio_window + 1)]


def progressive_sampling(unet_sd, cfg, sched, vq_sd, fs, audio_sd, c1, xid, xmasks, audio_feats, S,
                         audio_window, x_T_frames, fixed_identity=False):
    """DDIMSampler.progressive_sampling, TF progressive_sampling_difftalk.py:245-319 (eta=0, scale=1).

    x_T_frames: (T,1,C,H,W) start noise per frame (the reference draws torch.randn per frame).
    fixed_identity=True keeps zid = xid for every frame (SURVEY §0 F2 mode (b)); False is the
    reference's autoregressive behaviour (zid <- generated latent, :316-317).
    """
    T = audio_feats.shape[0]
    zid = xid.clone()
    frames = []
    for f in range(T):
        idx = audio_window_indices(f, T, audio_window)
        c2 = audio_temporal_attention(audio_sd, audio_feats[idx].unsqueeze(0))
        c12 = torch.cat([c1, c2], dim=2)
        c3 = encode_first_stage(vq_sd, fs, xmasks[f].unsqueeze(0))
        c34 = torch.cat([c3, zid], dim=1)
        img = ddim_sample(unet_sd, cfg, sched, S, x_T_frames[f], cond=c12, c_concat=c34, eta=0.0)
        frames.append(img)
        if not fixed_identity:
            zid = img.clone()
    return frames
