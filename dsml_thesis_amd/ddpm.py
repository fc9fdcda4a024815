"""LatentDiffusion facade: drop-in for the `pl.LightningModule` the reference's samplers and scripts drive
  face_reenactment/ldm/models/diffusion/ddpm.py:44-420 (DDPM), :423-1394 (LatentDiffusion), :1397-1423 (DiffusionWrapper)
  talking_face/ldm/models/diffusion/ddpm2cond.py:430-1297, :1300-1315 (two-condition variant)
  face_reenactment/ldm/modules/ema.py:5-76 (LitEma)

Sampling surface: schedules, `apply_model`, `p_sample_loop`/`sample`, `sample_log`, `log_images`,
`decode_first_stage`/`encode_first_stage`, `q_sample`, `ema_scope`, state-dict key layout (`model.diffusion_model.*`,
`model_ema.*`, `first_stage_model.*`, `cond_stage_model*.*`).  Training surface: `p_losses`,
`forward`, `training_step` (manual optimisation on the HIP training engine), `configure_optimizers`.
The class derives from pytorch_lightning.LightningModule when Lightning is installed, else nn.Module.
"""
from contextlib import contextmanager

import numpy as np
import torch
import torch.nn as nn

from . import lib as L
from . import schedule as S_
from .unet import rerun_if_flags_tripped
from .util import instantiate_from_config

try:  # north_star: keep the pl.LightningModule surface when Lightning exists
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # offline image: Lightning absent
    class _Base(nn.Module):
        @property
        def device(self):
            for p in self.parameters():
                return p.device
            for b in self.buffers():
                return b.device
            return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass


class LitEma(nn.Module):
    """EMA shadow of a module's parameters, stored as buffers whose names are the parameter names with the dots
    removed (the checkpoint layout of ema.py:16-21, so `model_ema.*` keys of reference checkpoints load as they are).

    `swap_in` / `swap_out` write THROUGH the parameters (`Parameter.copy_`, not `.data.copy_`): that bumps the
    parameters' version counters, which is what tells the packed-weight caches of the HIP models (UNetModel,
    the encoders, the autoencoder) that their kernel-layout copies are stale."""

    def __init__(self, model, decay=0.9999, use_num_upates=True):
        super().__init__()
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.register_buffer("decay", torch.tensor(decay, dtype=torch.float32))
        self.register_buffer("num_updates", torch.tensor(0 if use_num_upates else -1, dtype=torch.int))
        self.m_name2s_name = {name: name.replace(".", "") for name, _ in model.named_parameters()}
        for name, p in model.named_parameters():
            self.register_buffer(self.m_name2s_name[name], p.detach().clone())
        self.collected_params = []

    def shadow_of(self, name):
        return getattr(self, self.m_name2s_name[name])

    @torch.no_grad()
    def copy_to(self, model):
        """ema.py:46-55: shadow -> parameters."""
        for name, p in model.named_parameters():
            p.copy_(self.shadow_of(name))

    @torch.no_grad()
    def store(self, parameters):
        """ema.py:57-64."""
        self.collected_params = [p.detach().clone() for p in parameters]

    @torch.no_grad()
    def restore(self, parameters):
        """ema.py:66-76."""
        for saved, p in zip(self.collected_params, parameters):
            p.copy_(saved)
        self.collected_params = []


class DiffusionWrapper(_Base):
    """ddpm.py:1397-1423 / ddpm2cond.py:1300-1315: routes c_concat / c_crossattn into the UNet.  The channel
    concat is not materialised: the UNet's first conv reads both tensors."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, "concat", "crossattn", "hybrid", "adm"]

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None):
        """ddpm.py:1404-1423.  None: the unconditional UNet, `diffusion_model(x, t)`; 'concat': channel concat only (the UNet's
        first convolution reads both tensors, nothing is materialised); 'crossattn' / 'hybrid': context (+ concat)."""
        if self.conditioning_key == "adm":                # ddpm.py:1417-1420: the conditioning IS the class label vector y
            return self.diffusion_model(x, t, y=c_crossattn[0])
        cat = None
        if c_concat is not None and self.conditioning_key in ("concat", "hybrid", "crossattn"):
            cat = c_concat[0] if len(c_concat) == 1 else torch.cat(c_concat, 1)
        if self.conditioning_key is None:
            return self.diffusion_model(x, t)
        if self.conditioning_key == "concat":
            return self.diffusion_model(x, t, c_concat=cat)
        cc = torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc, c_concat=cat)


class LatentDiffusion(_Base):
    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="image",
                 cond_stage_trainable=False, concat_mode=True, cond_stage_forward=None, conditioning_key=None,
                 scale_factor=1.0, scale_by_std=False, unet_config=None, timesteps=1000, beta_schedule="linear",
                 loss_type="l2", ckpt_path=None, ignore_keys=[], load_only_unet=False, monitor="val/loss",
                 use_ema=True, first_stage_key="image", image_size=256, channels=3, log_every_t=100,
                 clip_denoised=True, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, given_betas=None,
                 original_elbo_weight=0., v_posterior=0., l_simple_weight=1., parameterization="eps",
                 scheduler_config=None, use_positional_encodings=False, learn_logvar=False, logvar_init=0., **kw):
        super().__init__()
        assert parameterization == "eps", "the shipped configs are eps-prediction"
        self.num_timesteps_cond = 1 if num_timesteps_cond is None else num_timesteps_cond
        assert self.num_timesteps_cond <= timesteps
        if self.num_timesteps_cond != 1:
            raise NotImplementedError("shorten_cond_schedule (num_timesteps_cond > 1) is unused by the shipped configs")
        if conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        if cond_stage_config == "__is_unconditional__":
            conditioning_key = None
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised = False                      # LatentDiffusion overrides DDPM's default, ddpm.py:463
        self.log_every_t, self.first_stage_key = log_every_t, first_stage_key
        self.image_size, self.channels = image_size, channels
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = use_ema
        if self.use_ema:
            self.model_ema = LitEma(self.model)
        self.v_posterior = v_posterior
        if monitor is not None:
            self.monitor = monitor
        self.scale_by_std = scale_by_std
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self.logvar = torch.full(fill_value=logvar_init, size=(self.num_timesteps,))
        self.concat_mode, self.cond_stage_trainable, self.cond_stage_key = concat_mode, cond_stage_trainable, cond_stage_key
        if not scale_by_std:
            self.scale_factor = scale_factor
        else:
            self.register_buffer("scale_factor", torch.tensor(scale_factor))
        self.instantiate_first_stage(first_stage_config)
        self.instantiate_cond_stage(cond_stage_config)
        self.cond_stage_forward = cond_stage_forward
        self.learn_logvar = learn_logvar
        if learn_logvar:
            raise NotImplementedError("learn_logvar=True is unused by the shipped configs")
        self.use_scheduler = scheduler_config is not None
        self.scheduler_config = scheduler_config
        self.restarted_from_ckpt = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)
            self.restarted_from_ckpt = True

    # ---- construction helpers ----------------------------------------------------------------------
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else S_.make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self.num_timesteps = int(np.asarray(betas).shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        for k, v in S_.schedule_buffers(betas, self.v_posterior).items():
            self.register_buffer(k, v)
        self.shorten_cond_schedule = False
        self._ddpm_tables = None

    def instantiate_first_stage(self, config):
        self.first_stage_model = instantiate_from_config(config).eval()
        for p in self.first_stage_model.parameters():
            p.requires_grad = False

    def instantiate_cond_stage(self, config):
        if config in ("__is_first_stage__", "__is_unconditional__"):
            self.cond_stage_model = self.first_stage_model if config == "__is_first_stage__" else None
            return
        self.cond_stage_model = instantiate_from_config(config)
        if not self.cond_stage_trainable:
            self.cond_stage_model.eval()

    def init_from_ckpt(self, path, ignore_keys=(), only_model=False):
        """ddpm.py:186-201: load a Lightning checkpoint (or a bare state dict), dropping keys by prefix."""
        state = torch.load(path, map_location="cpu")
        state = state["state_dict"] if "state_dict" in state else state
        dropped = [k for k in state if any(k.startswith(prefix) for prefix in ignore_keys)]
        for k in dropped:
            state.pop(k)
        target = self.model if only_model else self
        result = target.load_state_dict(state, strict=False)
        print(f"init_from_ckpt: {path}: dropped {len(dropped)} keys, {len(result.missing_keys)} missing, "
              f"{len(result.unexpected_keys)} unexpected")
        return result

    @contextmanager
    def ema_scope(self, context=None):
        """ddpm.py:171-184: sample with the EMA weights, then put the training weights back.  Every reference sampling
        script runs inside this scope.  The swap writes through the parameters (LitEma.copy_to / restore) as the reference's
        does; the UNet's kernel-layout copies of either weight set are packed once and swapped by name (see below)."""
        swapped = bool(self.use_ema)
        unet = self.model.diffusion_model
        if swapped:
            # name both weight sets for the UNet (UNetModel.adopt_weights): the kernel-layout copies, launch programs and captured
            # graphs of each are built once and swapped by reference afterwards -- the copies below still write through the
            # parameters, so everything that reads them (state_dict, the trainer) sees what the reference's scope shows
            sig, cur = unet._signature(), getattr(unet, "_weight_token", None)
            packed_is_current = unet._packed is not None and unet._pack_sig == sig
            if packed_is_current and cur is not None and cur[0] == "train":
                train_tok = cur                          # the set in use is the (unchanged) training set an earlier scope named
            else:
                train_tok = ("train", sig)               # new training weights (first scope, or an optimizer step since)
                if packed_is_current:
                    unet._weight_token = train_tok
            ema_tok = ("ema", tuple((b.data_ptr(), b._version) for b in self.model_ema.buffers()))
            self.model_ema.store(self.model.parameters())
            self.model_ema.copy_to(self.model)
            if next(unet.parameters()).is_cuda:
                unet.adopt_weights(ema_tok)
            if context is not None:
                print(f"{context}: Switched to EMA weights")
        try:
            yield None
        finally:
            if swapped:
                self.model_ema.restore(self.model.parameters())
                if next(unet.parameters()).is_cuda:
                    unet.adopt_weights(train_tok)
                if context is not None:
                    print(f"{context}: Restored training weights")

    # ---- model evaluation ----------------------------------------------------------------------------
    @torch.no_grad()
    def apply_model(self, x_noisy, t, cond, cond_concat=None, return_ids=False):
        """ddpm.py:893-994 (`apply_model(x,t,c)`) and ddpm2cond.py:911-945 (`apply_model(x,t,c12,c34)`)."""
        if cond is None:                                 # unconditional LDM (conditioning_key None)
            kwargs = {}
        elif isinstance(cond, dict):
            kwargs = dict(cond)
        else:
            if not isinstance(cond, list):
                cond = [cond]
            key = "c_concat" if self.model.conditioning_key == "concat" else "c_crossattn"
            kwargs = {key: cond}
        if cond_concat is not None:
            if isinstance(cond_concat, dict):
                kwargs.update(cond_concat)
            else:
                kwargs["c_concat"] = cond_concat if isinstance(cond_concat, list) else [cond_concat]
        return self.model(x_noisy, t, **kwargs)

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:230-233."""
        noise = torch.randn_like(x_start) if noise is None else noise
        sh = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
        return (self.sqrt_alphas_cumprod.gather(-1, t).reshape(sh) * x_start +
                self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(sh) * noise)

    def get_learned_conditioning(self, c):
        if self.cond_stage_forward is None:
            if hasattr(self.cond_stage_model, "encode") and callable(self.cond_stage_model.encode):
                return self.cond_stage_model.encode(c)
            return self.cond_stage_model(c)
        return getattr(self.cond_stage_model, self.cond_stage_forward)(c)

    # ---- first stage ---------------------------------------------------------------------------------
    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """ddpm.py:706-764 -> VQModelInterface.decode."""
        if predict_cids:
            raise NotImplementedError("decode_first_stage(predict_cids=True) is unused on the sampling path")
        if hasattr(self, "split_input_params"):
            raise NotImplementedError("patch-wise (split_input_params) decoding is unused by the shipped configs")
        z = 1. / self.scale_factor * z
        return self.first_stage_model.decode(z, force_not_quantize=force_not_quantize)

    @torch.no_grad()
    def encode_first_stage(self, x):
        """ddpm.py:826-864 -> VQModelInterface.encode (no quantisation)."""
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior):
        return self.scale_factor * encoder_posterior

    # ---- ancestral sampler -----------------------------------------------------------------------------
    def _ddpm_device_tables(self):
        if self._ddpm_tables is None or self._ddpm_tables[0].device != self.betas.device:
            tab = torch.stack([self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod,
                               self.posterior_mean_coef1, self.posterior_mean_coef2], 1).contiguous()
            self._ddpm_tables = (tab, self.posterior_log_variance_clipped.contiguous())
        return self._ddpm_tables

    # ---- the posterior pieces with the reference's names and return values (ddpm.py:203-228,1049-1078).  The sampling loops do
    # not go through them -- one step is the UNet program + ldmk_ddpm_step -- they are here for scripts that call them directly:
    # elementwise device work on the registered schedule buffers, `extract_into_tensor` (util.py:96-99) as a gather.
    @staticmethod
    def _extract(a, t, x):
        return a.gather(-1, t.to(torch.int64)).reshape((t.shape[0],) + (1,) * (x.dim() - 1))

    def q_mean_variance(self, x_start, t):
        """q(x_t | x_0): (mean, variance, log_variance), ddpm.py:203-213."""
        ex = lambda a: self._extract(a, t, x_start)
        return ex(self.sqrt_alphas_cumprod) * x_start, ex(1.0 - self.alphas_cumprod), ex(self.log_one_minus_alphas_cumprod)

    def predict_start_from_noise(self, x_t, t, noise):
        """ddpm.py:215-219."""
        return self._extract(self.sqrt_recip_alphas_cumprod, t, x_t) * x_t - self._extract(self.sqrt_recipm1_alphas_cumprod, t, x_t) * noise

    def q_posterior(self, x_start, x_t, t):
        """q(x_{t-1} | x_t, x_0): (mean, variance, clipped log variance), ddpm.py:221-228."""
        ex = lambda a: self._extract(a, t, x_t)
        mean = ex(self.posterior_mean_coef1) * x_start + ex(self.posterior_mean_coef2) * x_t
        return mean, ex(self.posterior_variance), ex(self.posterior_log_variance_clipped)

    @torch.no_grad()
    def p_mean_variance(self, x, c, t, clip_denoised: bool, return_codebook_ids=False, quantize_denoised=False, return_x0=False,
                        score_corrector=None, corrector_kwargs=None):
        """ddpm.py:1049-1078: (model_mean, posterior_variance, posterior_log_variance[, x_recon]) from one UNet evaluation."""
        if return_codebook_ids:
            raise NotImplementedError("p_mean_variance: return_codebook_ids needs the id-predictor head (n_embed), not built")
        model_out = self.apply_model(x, t, c)
        if score_corrector is not None:
            assert self.parameterization == "eps"
            model_out = score_corrector.modify_score(self, model_out, x, t, c, **(corrector_kwargs or {}))
        if self.parameterization == "eps":
            x_recon = self.predict_start_from_noise(x, t=t, noise=model_out)
        elif self.parameterization == "x0":
            x_recon = model_out
        else:
            raise NotImplementedError()
        if clip_denoised:
            x_recon = x_recon.clamp(-1., 1.)
        if quantize_denoised:
            x_recon = self.first_stage_model.quantize(x_recon)[0]
        out = self.q_posterior(x_start=x_recon, x_t=x, t=t)
        return out + (x_recon,) if return_x0 else out

    @torch.no_grad()
    def p_sample(self, x, c, t, clip_denoised=False, repeat_noise=False, return_codebook_ids=False,
                 quantize_denoised=False, return_x0=False, temperature=1., noise_dropout=0., score_corrector=None,
                 corrector_kwargs=None, noise=None):
        """ddpm.py:1080-1109 (with p_mean_variance :1049-1078, q_posterior :221-228) as one fused update kernel.  The
        options no shipped script sets (clip / quantise the x0 estimate, a score corrector, return_x0) take the same
        arithmetic as a few elementwise device ops around the UNet evaluation instead."""
        if return_codebook_ids:
            raise DeprecationWarning("Support dropped.")                    # ddpm.py:1092
        eps = self.apply_model(x, t, c)
        if noise is None:
            noise = torch.randn_like(x)
        noise = noise * temperature
        if noise_dropout > 0.:
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        t = t.to(torch.int64)
        if not (clip_denoised or quantize_denoised or score_corrector is not None or return_x0):
            tab, logvar = self._ddpm_device_tables()
            out = torch.empty_like(x)
            L.call("ldmk_ddpm_step", x.contiguous().data_ptr(), eps.data_ptr(), noise.contiguous().data_ptr(), tab.data_ptr(),
                   logvar.data_ptr(), t.contiguous().data_ptr(), out.data_ptr(), x[0].numel(), x.shape[0],
                   torch.cuda.current_stream().cuda_stream)
            return out
        if score_corrector is not None:
            assert self.parameterization == "eps"
            eps = score_corrector.modify_score(self, eps, x, t, c, **(corrector_kwargs or {}))
        ex = lambda a: a.gather(-1, t).view(-1, 1, 1, 1)                    # extract_into_tensor, util.py:96-99
        x_recon = ex(self.sqrt_recip_alphas_cumprod) * x - ex(self.sqrt_recipm1_alphas_cumprod) * eps      # :215-219
        if clip_denoised:
            x_recon = x_recon.clamp(-1., 1.)
        if quantize_denoised:
            x_recon = self.first_stage_model.quantize(x_recon)[0]
        mean = ex(self.posterior_mean_coef1) * x_recon + ex(self.posterior_mean_coef2) * x                # q_posterior :221-228
        nonzero = (1 - (t == 0).float()).view(-1, 1, 1, 1)
        out = mean + nonzero * (0.5 * ex(self.posterior_log_variance_clipped)).exp() * noise
        return (out, x_recon) if return_x0 else out

    @torch.no_grad()
    @rerun_if_flags_tripped(lambda self: self.model.diffusion_model)
    def p_sample_loop(self, cond, shape, return_intermediates=False, x_T=None, verbose=True, callback=None,
                      timesteps=None, quantize_denoised=False, mask=None, x0=None, img_callback=None, start_T=None,
                      log_every_t=None, noise=None, use_graph=False, mask_noise=None):
        """ddpm.py:1167-1216.  Device-resident loop: the latent lives in the UNet program's input buffer, the
        per-sample timestep vector is rewritten by a kernel, one step = UNet program + fused posterior update, and
        (use_graph=True) the step is captured once in a hipGraph and replayed T times.
        `noise`: optional per-step noise list (parity with a seeded reference run)."""
        from .engine import GraphedProgram
        log_every_t = log_every_t or self.log_every_t
        dev = self.betas.device
        b = shape[0]
        img0 = torch.randn(shape, device=dev) if x_T is None else x_T.to(dev, torch.float32)
        timesteps = self.num_timesteps if timesteps is None else timesteps
        if start_T is not None:
            timesteps = min(timesteps, start_T)
        if mask is not None or quantize_denoised or self.clip_denoised:
            # the options no shipped script sets (ddpm.py:1199-1208): step by step through p_sample, eager launches;
            # `mask_noise` (a per-step list, like `noise`) pins q_sample's draw for parity with a seeded reference run
            if mask is not None:
                assert x0 is not None and x0.shape[2:3] == mask.shape[2:3]
                mask, x0 = mask.to(dev, torch.float32), x0.to(dev, torch.float32)
            img, intermediates = img0, [img0]
            for k, i in enumerate(reversed(range(0, timesteps))):
                ts = torch.full((b,), i, device=dev, dtype=torch.long)
                img = self.p_sample(img, cond, ts, clip_denoised=self.clip_denoised, quantize_denoised=quantize_denoised,
                                    noise=None if noise is None else noise[k].to(dev))
                if mask is not None:
                    img_orig = self.q_sample(x0, ts, noise=None if mask_noise is None else mask_noise[k].to(dev))
                    img = img_orig * mask + (1. - mask) * img
                if i % log_every_t == 0 or i == timesteps - 1:
                    intermediates.append(img)
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(img, i)
            return (img, intermediates) if return_intermediates else img
        if self.model.conditioning_key == "adm":          # labels in any form apply_model accepts: y, [y], {"c_crossattn": [y]}
            ctx = cond.get("c_crossattn") if isinstance(cond, dict) else cond
            ctx, cat = (ctx[0] if isinstance(ctx, (list, tuple)) else ctx), None
        elif isinstance(cond, dict):
            ctx = cond.get("c_crossattn")
            cat = cond.get("c_concat")
            ctx = torch.cat(ctx, 1) if isinstance(ctx, (list, tuple)) else ctx
            cat = torch.cat(cat, 1) if isinstance(cat, (list, tuple)) else cat
        elif self.model.conditioning_key == "concat":
            ctx, cat = None, (torch.cat(cond, 1) if isinstance(cond, (list, tuple)) else cond)
        else:
            ctx, cat = (torch.cat(cond, 1) if isinstance(cond, (list, tuple)) else cond), None      # (None: unconditional)
        unet = self.model.diffusion_model
        y = None
        if self.model.conditioning_key == "adm":          # class labels: rows of the label embedding, constant over the run (ddpm.py:1417-1419)
            y, ctx = ctx, None
            assert y is not None and y.dim() == 1 and cat is None, "conditioning_key 'adm': cond = class labels (B,)"
        ncat = 0 if cat is None else cat.shape[1]
        L_ctx = 0 if ctx is None else ctx.shape[1]
        pg = unet.program(b, shape[2], shape[3], L_ctx, ncat)
        if L_ctx:
            pg.inputs["context"].copy_(ctx.reshape(b * L_ctx, -1))
        if ncat:
            pg.inputs["c_concat"].copy_(cat)
        if y is not None:
            pg.inputs["y_emb"].copy_(unet.label_emb.weight.detach().float()[y.to(dev, torch.int64)])
        pg.ctx_program.run()
        x_buf, t_buf, eps = pg.inputs["x"], pg.inputs["t"], pg.outputs["eps"]
        tab, logvar = self._ddpm_device_tables()
        key = (timesteps, bool(use_graph), noise is None)
        loops = pg.__dict__.setdefault("_ddpm_loops", {})       # (the state lives on the program and dies with it, like ddim.py's)
        st = loops.get(key)
        if st is None:
            st = dict(nz=torch.empty_like(img0), idx=torch.zeros(1, dtype=torch.int32, device=dev),
                      tt=torch.arange(self.num_timesteps, dtype=torch.int64, device=dev), graph=None)
            loops[key] = st
        nz, idx, tt = st["nz"], st["idx"], st["tt"]
        per = img0[0].numel()
        lib = pg.lib

        def reset():
            x_buf.copy_(img0)
            idx.fill_(timesteps - 1)
            t_buf.fill_(timesteps - 1)

        def one_step(draw=True):
            if draw:
                nz.normal_()
            pg.run()
            stream = torch.cuda.current_stream().cuda_stream
            L.check(lib.ldmk_ddpm_step(x_buf.data_ptr(), eps.data_ptr(), nz.data_ptr(), tab.data_ptr(), logvar.data_ptr(),
                                       t_buf.data_ptr(), x_buf.data_ptr(), per, b, stream), "ldmk_ddpm_step")
            L.check(lib.ldmk_advance_timestep(idx.data_ptr(), tt.data_ptr(), t_buf.data_ptr(), b, 1, self.num_timesteps,
                                              stream), "ldmk_advance_timestep")

        reset()
        step = one_step
        if noise is not None:
            step = lambda: one_step(draw=False)
        elif use_graph:
            if st["graph"] is None:
                st["graph"] = GraphedProgram(one_step)
                reset()
            step = st["graph"].replay
        intermediates = [img0]
        for k, i in enumerate(reversed(range(0, timesteps))):
            if noise is not None:
                nz.copy_(noise[k])
            step()
            if i % log_every_t == 0 or i == timesteps - 1:
                intermediates.append(x_buf.clone())
            if callback:
                callback(i)
            if img_callback:
                img_callback(x_buf, i)
        img = x_buf.clone()
        return (img, intermediates) if return_intermediates else img

    @torch.no_grad()
    def sample(self, cond, batch_size=16, return_intermediates=False, x_T=None, verbose=True, timesteps=None,
               quantize_denoised=False, mask=None, x0=None, shape=None, **kwargs):
        """ddpm.py:1218-1234."""
        if shape is None:
            shape = (batch_size, self.channels, self.image_size, self.image_size)
        if cond is not None:
            if isinstance(cond, dict):
                cond = {k: (v[:batch_size] if not isinstance(v, list) else [x[:batch_size] for x in v])
                        for k, v in cond.items()}
            else:
                cond = [c[:batch_size] for c in cond] if isinstance(cond, list) else cond[:batch_size]
        return self.p_sample_loop(cond, shape, return_intermediates=return_intermediates, x_T=x_T, verbose=verbose,
                                  timesteps=timesteps, quantize_denoised=quantize_denoised, mask=mask, x0=x0, **kwargs)

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        """ddpm.py:1236-1249 (what ImageLogger drives during training)."""
        if ddim:
            from .ddim import DDIMSampler
            shape = (self.channels, self.image_size, self.image_size)
            return DDIMSampler(self).sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)
        return self.sample(cond=cond, batch_size=batch_size, return_intermediates=True, **kwargs)

    # ---- training surface -------------------------------------------------------------
    automatic_optimization = False      # Lightning: the step below runs its own backward + optimizer (no autograd)
    # arithmetic of the training step's GEMMs: "f32" (parity path) or "bf16" (BASELINE configs[4]: bf16 matrix-core
    # products, fp32 accumulation / master weights / AdamW / EMA).  Set it before the first training call, e.g.
    # `model.train_compute = "bf16"` -- the role `precision=bf16` plays for the reference's Lightning Trainer (main.py)
    train_compute = "f32"

    def trainer(self):
        """The HIP training engine for the UNet (forward + hand-written backward, flat packed parameters)."""
        if getattr(self, "_trainer", None) is None:
            from .train import UNetTrainer
            self._trainer = UNetTrainer(self.model.diffusion_model, compute=self.train_compute)
            self._ema_flat = None
            if self.use_ema:
                # the shadow starts from the `model_ema` buffers (what a checkpoint restored), not from the live
                # weights: resuming keeps the stored average like the reference's LitEma does (ema.py:25-44)
                pre = "diffusion_model."
                ema_sd = {k[len(pre):]: self.model_ema.shadow_of(k) for k in self.model_ema.m_name2s_name
                          if k.startswith(pre)}
                self._ema_flat = self._trainer.pack_reference_state(ema_sd)
            if not hasattr(self, "_cond_opt"):       # (configure_optimizers may already have created the one Lightning owns)
                self._cond_opt = None
            pending = self.__dict__.pop("_pending_training_state", None)
            if pending is not None:                  # a Lightning checkpoint written by on_save_checkpoint
                self.load_training_state(pending)
        return self._trainer

    def training_state(self):
        """Everything a faithful resume needs beyond the module's own state dict: packed weights + AdamW moments + step
        (UNetTrainer.optimizer_state), the packed EMA shadow, and the conditioner optimisers."""
        tr = self.trainer()
        st = dict(unet=tr.optimizer_state(), ema_flat=None if self._ema_flat is None else self._ema_flat.clone())
        for key in ("_cond_opt", "_audio_opt"):
            opt = getattr(self, key, None)
            st[key] = None if opt is None else opt.state_dict()
        return st

    def load_training_state(self, st, lr=1e-4, weight_decay=1e-2):
        tr = self.trainer()
        tr.load_optimizer_state(st["unet"])
        if st.get("ema_flat") is not None and self.use_ema:
            self._ema_flat.copy_(st["ema_flat"])
        owners = {"_cond_opt": getattr(self, "cond_stage_model_1", None) or self.cond_stage_model,
                  "_audio_opt": getattr(self, "cond_stage_model_2", None)}
        for key, owner in owners.items():
            if st.get(key) is not None and owner is not None:
                opt = torch.optim.AdamW(owner.parameters(), lr=lr, weight_decay=weight_decay)
                opt.load_state_dict(st[key])
                setattr(self, key, opt)

    def _context_tensor(self, cond):
        if isinstance(cond, dict):
            cond = cond.get("c_crossattn", cond)
        if isinstance(cond, (list, tuple)):
            cond = torch.cat(list(cond), 1)
        return cond

    def p_losses(self, x_start, cond, t, noise=None, c_concat=None, reduce_world=1):
        """ddpm.py:1014-1047 for the shipped settings (eps-prediction, l2, l_simple_weight 1, fixed logvar 0,
        original_elbo_weight 0): loss = mean((eps_theta(q_sample(x0, t, noise), t, cond) - noise)^2).
        Runs forward AND backward on the HIP kernels; gradients are left in `self.trainer().P.grad` (UNet, packed
        layout) and `self.trainer().dctx` (w.r.t. the context tokens).  Returns (loss, loss_dict) like the reference;
        the logged-only `loss_vlb` entry is not computed."""
        if self.model.conditioning_key not in ("crossattn", "hybrid"):
            raise NotImplementedError("p_losses: cross-attention conditioned UNets only")
        tr = self.trainer()
        noise = torch.randn_like(x_start) if noise is None else noise
        ctx = self._context_tensor(cond)
        if isinstance(c_concat, (list, tuple)):
            c_concat = torch.cat(list(c_concat), 1)
        loss = tr.p_losses(x_start.float(), ctx, t, noise.float(), self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod,
                           c_concat=c_concat, reduce_world=reduce_world)
        prefix = "train" if self.training else "val"
        return loss, {f"{prefix}_loss_simple": loss, f"{prefix}_loss": loss}

    def forward(self, x, c, *args, **kwargs):
        """ddpm.py:866-877: draw t, embed the condition when the conditioner is trainable, p_losses."""
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=x.device).long()
        if self.cond_stage_trainable:
            c = self.get_learned_conditioning(c)
        return self.p_losses(x, c, t, *args, **kwargs)

    def training_step_latents(self, z, cond_batch, lr, t=None, noise=None, world_size=1, weight_decay=1e-2):
        """One optimisation step on already-encoded latents z (what `shared_step` produces from a batch,
        ddpm.py:879-881): loss + backward on the HIP kernels, gradient all-reduce when world_size > 1 (DDP,
        main.py:532), AdamW on the UNet (ldmk_adamw over the flat buffer) and on the conditioner (torch, a few KB),
        then the EMA update of on_train_batch_end (ddpm.py:396-398)."""
        tr = self.trainer()
        t = torch.randint(0, self.num_timesteps, (z.shape[0],), device=z.device).long() if t is None else t
        c = cond_batch
        if self.cond_stage_trainable:
            with torch.enable_grad():
                try:
                    c = self.cond_stage_model(cond_batch, training=self.training)
                except TypeError:
                    c = self.get_learned_conditioning(cond_batch)
        ctx = self._context_tensor(c)
        loss, loss_dict = self.p_losses(z, ctx.detach(), t, noise, reduce_world=world_size)   # bucketed all-reduce inside
        tr.adamw_step(lr, weight_decay=weight_decay)
        if self.cond_stage_trainable and ctx.requires_grad:
            if self._cond_opt is None:
                self._cond_opt = torch.optim.AdamW(self.cond_stage_model.parameters(), lr=lr, weight_decay=weight_decay)
            for grp in self._cond_opt.param_groups:
                grp["lr"] = lr
            self._cond_opt.zero_grad(set_to_none=True)
            dctx = tr.dctx.view_as(ctx)
            if world_size > 1:
                import torch.distributed as dist
                dist.all_reduce(dctx)
                dctx = dctx / world_size
            ctx.backward(dctx)
            self._cond_opt.step()
        if self.use_ema:
            decay = float(self.model_ema.decay)
            if int(self.model_ema.num_updates) >= 0:                                    # ema.py:28-31
                self.model_ema.num_updates += 1
                n_up = int(self.model_ema.num_updates)
                decay = min(decay, (1 + n_up) / (10 + n_up))
            tr.ema_update(self._ema_flat, decay)
        return loss, loss_dict

    def training_step(self, batch, batch_idx=0):
        """ddpm.py:341-358 with manual optimisation: batch -> (latents, condition) -> one full step."""
        z, c = self.get_input(batch, self.first_stage_key)
        lr = float(getattr(self, "learning_rate", 1e-4))
        if getattr(self, "lr_schedule", None) is not None:
            lr *= float(self.lr_schedule(self.trainer().P.step))
        loss, loss_dict = self.training_step_latents(z, c, lr)
        self.log_dict(loss_dict, prog_bar=True, logger=True, on_step=True, on_epoch=True)
        return loss

    shared_step = training_step

    def get_input(self, batch, k):
        """ddpm.py:667-704 for the shipped keys: images (b,h,w,c) in [-1,1] -> first-stage latents; the raw batch is
        the condition when the conditioner is trainable (it embeds `batch[cond_stage_key]` itself)."""
        x = batch[k]
        if x.dim() == 3:
            x = x[..., None]
        x = x.permute(0, 3, 1, 2).contiguous().float().to(self.device)
        z = self.get_first_stage_encoding(self.encode_first_stage(x)).detach()
        return z, batch

    def configure_optimizers(self):
        """ddpm.py:1363-1385.  The UNet's AdamW runs inside the HIP training engine (`ldmk_adamw` over the flat packed
        buffer, driven by `training_step`), so what Lightning receives here is the optimiser of the parameters that
        stay under torch autograd: the trainable conditioner.  `learning_rate` is read per step by `training_step`; a
        `scheduler_config` is instantiated and exposed as `lr_schedule` (step -> multiplier), which `training_step`
        applies to both optimisers."""
        lr = float(getattr(self, "learning_rate", 1e-4))
        self.lr_schedule = None
        if self.use_scheduler:
            assert "target" in self.scheduler_config
            self.lr_schedule = instantiate_from_config(self.scheduler_config).schedule
        owner = self.cond_stage_model if self.cond_stage_model is not None else getattr(self, "cond_stage_model_1", None)
        if not self.cond_stage_trainable or owner is None:
            return None
        print(f"{self.__class__.__name__}: Also optimizing conditioner params!")
        # the very object training_step_latents steps (it only creates one when none exists), so that what Lightning
        # checkpoints are the moments that were actually updated
        self._cond_opt = torch.optim.AdamW(owner.parameters(), lr=lr)
        return self._cond_opt

    @torch.no_grad()
    def log_images(self, batch, N=8, n_row=4, sample=True, ddim_steps=200, ddim_eta=1., return_keys=None,
                   quantize_denoised=False, inpaint=False, plot_denoise_rows=False, plot_progressive_rows=False,
                   plot_diffusion_rows=False, **kwargs):
        """ddpm.py:1253-1361, the hook ImageLogger calls during training: inputs, first-stage reconstruction, and
        samples drawn under `ema_scope` (DDIM when `ddim_steps` is set), decoded.  The diagnostic rows the shipped
        logger settings leave off (denoise / progressive / diffusion rows, inpainting, quantised x0) raise."""
        if quantize_denoised or inpaint or plot_denoise_rows or plot_progressive_rows or plot_diffusion_rows:
            raise NotImplementedError("log_images: only inputs / reconstruction / samples are produced")
        # during training the live weights and the EMA live in the training engine's flat buffers: bring the module tree
        # (what ema_scope / the samplers read) up to date first, or the logged samples would show the step-0 weights
        self._sync_if_training()
        x = batch[self.first_stage_key]
        x = (x[..., None] if x.dim() == 3 else x)[:N].permute(0, 3, 1, 2).contiguous().float().to(self.device)
        z = self.get_first_stage_encoding(self.encode_first_stage(x))
        log = {"inputs": x, "reconstruction": self.decode_first_stage(z)}
        if sample:
            c = self._log_conditioning(batch, x.shape[0])
            with self.ema_scope("Plotting"):
                samples, _ = self.sample_log(cond=c, batch_size=x.shape[0], ddim=ddim_steps is not None,
                                             ddim_steps=ddim_steps, eta=ddim_eta)
            log["samples"] = self.decode_first_stage(samples)
        if return_keys and any(k in log for k in return_keys):
            return {k: log[k] for k in return_keys if k in log}
        return log

    def _log_conditioning(self, batch, n):
        """The condition `log_images` samples with (ddpm.py:1262-1266: get_input(..., return_first_stage_outputs=True))."""
        cond = batch[self.cond_stage_key]
        cond = {self.cond_stage_key: cond[:n]} if not isinstance(cond, dict) else cond
        cond = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in cond.items()}
        return self.get_learned_conditioning(cond)

    def _sync_if_training(self):
        if getattr(self, "_trainer", None) is not None:
            self.sync_trained_weights()

    def on_save_checkpoint(self, checkpoint):
        """Lightning hook: a checkpoint taken mid-training must hold the trained weights and EMA (they live in the training
        engine's flat buffers between steps), plus the engine's optimiser state for a faithful resume."""
        if getattr(self, "_trainer", None) is not None:
            self.sync_trained_weights()
            if isinstance(checkpoint, dict):
                if "state_dict" in checkpoint:
                    checkpoint["state_dict"] = {k: v.detach().clone() for k, v in super().state_dict().items()}
                checkpoint["ldmk_training_state"] = self.training_state()

    def on_load_checkpoint(self, checkpoint):
        st = checkpoint.get("ldmk_training_state") if isinstance(checkpoint, dict) else None
        if st is not None:
            self._pending_training_state = st

    def state_dict(self, *args, **kwargs):
        """nn.Module.state_dict, after copying the trained weights / EMA out of the training engine when one exists."""
        self._sync_if_training()
        return super().state_dict(*args, **kwargs)

    def sync_trained_weights(self):
        """Copy the trained (and EMA) weights from the training engine back into the nn.Module tree, so that
        `ema_scope()`, `state_dict()` and the samplers see them."""
        tr = self.trainer()
        tr.sync_to_module()
        if self.use_ema:
            sd = tr.state_dict_reference(self._ema_flat)
            shadow = dict(self.model_ema.named_buffers())
            for k, v in sd.items():
                name = self.model_ema.m_name2s_name.get("diffusion_model." + k)
                if name is not None:
                    shadow[name].copy_(v)


class LatentDiffusion2Cond(LatentDiffusion):
    """talking_face ddpm2cond.LatentDiffusion: class-label + audio cross-attention tokens concatenated on the
    feature axis, masked-frame + identity latents concatenated on the channel axis (ddpm2cond.py:430-1297)."""

    def __init__(self, first_stage_config, cond_stage_config_1, cond_stage_config_2, cond_stage_key_1="class_label",
                 cond_stage_key_2="audio", **kwargs):
        super().__init__(first_stage_config, cond_stage_config_1, cond_stage_key=cond_stage_key_1, **kwargs)
        self.cond_stage_key_1, self.cond_stage_key_2 = cond_stage_key_1, cond_stage_key_2
        self.cond_stage_model_1 = self.cond_stage_model
        self.cond_stage_model = None
        self.cond_stage_model_2 = instantiate_from_config(cond_stage_config_2)

    def _log_conditioning(self, batch, n):
        """ddpm2cond.py log_images: c12 = cat([class embedding, audio-window feature], 2) as the cross-attention token,
        c34 = cat([encode(masked frame), encode(identity)], 1) on the channel axis (get_input, ddpm2cond.py:676-741)."""
        def img(k):
            v = batch[k][:n]
            return (v[..., None] if v.dim() == 3 else v).permute(0, 3, 1, 2).contiguous().float().to(self.device)
        c1 = self.cond_stage_model_1({self.cond_stage_key_1: batch[self.cond_stage_key_1][:n].to(self.device)})
        c2 = self.cond_stage_model_2(batch[self.cond_stage_key_2][:n].float().to(self.device))
        z_mask = self.get_first_stage_encoding(self.encode_first_stage(img("masked_image")))
        z_id = self.get_first_stage_encoding(self.encode_first_stage(img("identity")))
        from .ddim import C12, C34
        return {C12: torch.cat([c1, c2], 2), C34: torch.cat([z_mask, z_id], 1)}

    def p_losses(self, x_start, cond12, cond34=None, t=None, noise=None):
        """ddpm2cond.py p_losses(x_start, cond12, cond34, t): cross-attention tokens + channel-concat latents."""
        return super().p_losses(x_start, cond12, t, noise=noise, c_concat=cond34)

    def training_step_latents(self, z, cond_batch, audio_feat, c34, lr, t=None, noise=None, world_size=1, weight_decay=1e-2,
                              audio_window=None):
        """One optimisation step of the talking-face model on encoded latents (ddpm2cond.py shared_step -> forward ->
        p_losses): c12 = cat([class embedding (B,1,256), audio feature (B,1,768)], 2) as cross-attention token, c34 =
        masked-frame + identity latents (B,6,h,w) concatenated on the channel axis.  The UNet and the class embedder
        are optimised; `audio_feat` is the (already pooled) output of cond_stage_model_2 -- its gradient is returned in
        the loss dict as `d_audio_feat`.  Pass `audio_window` (B,T,768) instead (audio_feat=None) to run and train
        cond_stage_model_2 (Conv1DTemporalAttention) as well: forward and backward on its fused HIP kernels."""
        tr = self.trainer()
        if audio_window is not None:
            audio_feat = self.cond_stage_model_2(audio_window)
        t = torch.randint(0, self.num_timesteps, (z.shape[0],), device=z.device).long() if t is None else t
        with torch.enable_grad():
            try:
                c1 = self.cond_stage_model_1(cond_batch, training=self.training)
            except TypeError:
                c1 = self.cond_stage_model_1(cond_batch)
        c12 = torch.cat([c1.detach(), audio_feat.detach().float()], 2)
        loss, loss_dict = self.p_losses(z, c12, c34, t, noise)
        if world_size > 1:
            tr.all_reduce_grads(world_size)
        tr.adamw_step(lr, weight_decay=weight_decay)
        dc12 = tr.dctx.view_as(c12)
        if self.cond_stage_trainable and c1.requires_grad:
            if getattr(self, "_cond_opt", None) is None:
                self._cond_opt = torch.optim.AdamW(self.cond_stage_model_1.parameters(), lr=lr, weight_decay=weight_decay)
            for grp in self._cond_opt.param_groups:
                grp["lr"] = lr
            self._cond_opt.zero_grad(set_to_none=True)
            c1.backward(dc12[..., :c1.shape[2]].contiguous())
            self._cond_opt.step()
        if self.use_ema:
            decay = float(self.model_ema.decay)
            if int(self.model_ema.num_updates) >= 0:
                self.model_ema.num_updates += 1
                n_up = int(self.model_ema.num_updates)
                decay = min(decay, (1 + n_up) / (10 + n_up))
            tr.ema_update(self._ema_flat, decay)
        d_audio = dc12[..., c1.shape[2]:]
        if audio_window is not None and self.cond_stage_trainable:
            from .encoders import audio_attention_backward
            audio_attention_backward(self.cond_stage_model_2, audio_window, d_audio)
            if getattr(self, "_audio_opt", None) is None:
                for p_ in self.cond_stage_model_2.parameters():
                    p_.requires_grad_(True)
                self._audio_opt = torch.optim.AdamW(self.cond_stage_model_2.parameters(), lr=lr, weight_decay=weight_decay)
            for grp in self._audio_opt.param_groups:
                grp["lr"] = lr
            self._audio_opt.step()
        loss_dict = dict(loss_dict, d_audio_feat=d_audio)
        return loss, loss_dict

    @torch.no_grad()
    def apply_model(self, x_noisy, t, cond12, cond34=None, return_ids=False):
        c12 = cond12 if isinstance(cond12, (list, dict)) else [cond12]
        c34 = cond34 if isinstance(cond34, (list, dict)) or cond34 is None else [cond34]
        kwargs = dict(c12) if isinstance(c12, dict) else {"c_crossattn": c12}
        if c34 is not None:
            kwargs.update(c34 if isinstance(c34, dict) else {"c_concat": c34})
        return self.model(x_noisy, t, **kwargs)
