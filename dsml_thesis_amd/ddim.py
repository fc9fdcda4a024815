"""DDIMSampler: drop-in for the reference samplers
  face_reenactment/ldm/models/diffusion/ddim.py:11-219      (one cross-attention condition, CFG by batch doubling)
  talking_face/ldm/models/diffusion/ddim2cond.py:11-308     (dict conditioning {'class_label_&_audio', 'motion_&_id'})
  talking_face/progressive_sampling_difftalk.py:245-319     (progressive_sampling, autoregressive identity)

Same call surface (`make_schedule`, `sample`, `ddim_sampling`, `p_sample_ddim`, `progressive_sampling`); the
loop itself is MI355X-native: the latent, the timestep vector, the step counter and the coefficient table stay
in device memory, one step = the UNet launch program + one fused update kernel, no host sync, and the step can
be captured once in a hipGraph and replayed S times.  Additions over the reference: `noise=` (pre-generated
per-step noise for eta>0 parity), `use_graph=`, `policy_batch=` (sharding-invariant tile choice) and
`fixed_identity=` for progressive sampling.
"""
import numpy as np
import torch

from . import lib as L
from . import schedule as S_
from .engine import GraphedProgram
from .unet import rerun_if_flags_tripped

C12, C34 = "class_label_&_audio", "motion_&_id"


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        # device-resident schedule tables, shared by every sampler of this model (captured graphs keep pointing at them, so a
        # second DDIMSampler(model) replays the step the first one captured instead of capturing it again)
        self._tables = model.__dict__.setdefault("_ddim_tables", {})

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True, strength=None):
        """`strength` (latent-manipulation scripts only, compute_latents.py:108-112): strength-scaled timesteps."""
        if strength is None:
            self.ddim_timesteps = S_.make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps)
        else:
            assert ddim_discretize == "uniform"
            self.ddim_timesteps = S_.make_ddim_timesteps_strength(ddim_num_steps, self.ddpm_num_timesteps, strength)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        sig, al, alp = S_.make_ddim_sampling_parameters(ac.cpu(), self.ddim_timesteps, ddim_eta)
        self.register_buffer("ddim_sigmas", torch.as_tensor(np.asarray(sig)))
        self.register_buffer("ddim_alphas", torch.as_tensor(al))
        self.ddim_alphas_prev = alp
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - al)
        # device-resident [S][4] coefficient table + timestep table for the update kernel
        dev = self.model.device
        key = (ddim_num_steps, float(ddim_eta), ddim_discretize, strength, str(dev))
        if key not in self._tables:     # persistent device tables: captured graphs keep pointing at them
            self._tables[key] = (torch.from_numpy(S_.ddim_step_table(ac.cpu(), self.ddim_timesteps, ddim_eta)).to(dev),
                                 torch.from_numpy(self.ddim_timesteps.astype(np.int64)).to(dev),
                                 torch.from_numpy(S_.ddim_inversion_table(ac.cpu(), self.ddim_timesteps)).to(dev))
        self._table, self._ts_table, self._inv_table = self._tables[key]
        self._sched_key = key
        self._eta = ddim_eta

    def _original_tables(self):
        """`use_original_steps` (ddim.py:127,133-134,183-186; ddim2cond.py:175-178): the update walks all `ddpm_num_timesteps` of the
        model's own schedule -- alphas_cumprod, alphas_cumprod_prev, sqrt_one_minus_alphas_cumprod and
        `ddim_sigmas_for_original_num_steps` = eta sqrt((1 - a_prev) / (1 - a) (1 - a / a_prev)), formed in float32 from the float32
        buffers like make_schedule does (ddim.py:50-53).  -> ([T][4] device table, arange(T) device, arange(T) host).
        (The face-reenactment copy reads the sigmas off `self.model`, where nothing defines them -- ddim.py:186 raises
        AttributeError; the talking-face copy, ddim2cond.py:178, reads its own buffer.  That working behaviour is what is built.)"""
        m, dev = self.model, self.model.device
        key = ("orig", float(self._eta), str(dev))
        if key not in self._tables:
            a = m.alphas_cumprod.to(dev, torch.float32)
            ap = m.alphas_cumprod_prev.to(dev, torch.float32)
            sig = self._eta * torch.sqrt((1 - ap) / (1 - a) * (1 - a / ap))
            tab = torch.stack([a, ap, sig.to(torch.float32), m.sqrt_one_minus_alphas_cumprod.to(dev, torch.float32)], 1).contiguous()
            self._tables[key] = (tab, torch.arange(self.ddpm_num_timesteps, dtype=torch.int64, device=dev), None)
        tab, ts, _ = self._tables[key]
        return tab, ts, np.arange(self.ddpm_num_timesteps)

    # ------------------------------------------------------------------------------------------
    def _is_adm(self):
        return getattr(self.model.model, "conditioning_key", None) == "adm"

    @staticmethod
    def _labels(cond):
        """conditioning_key 'adm': the class-label vector y (B,) out of whatever form apply_model accepts (ddpm.py:893-994 wraps a
        tensor as c_crossattn = [y]; DiffusionWrapper takes c_crossattn[0], :1417-1419)."""
        if isinstance(cond, dict):
            cond = cond.get("c_crossattn")
        if isinstance(cond, (list, tuple)):
            cond = cond[0]
        return cond

    def _split_cond(self, cond):
        """-> (crossattn context (B,L,D), c_concat (B,C,H,W) or None); 'adm': (class labels (B,), None)."""
        if self._is_adm():
            return self._labels(cond), None
        if isinstance(cond, dict):
            if C12 in cond:                      # TF sampler conditioning, ddim2cond.py:165
                return cond[C12], cond[C34]
            cc = cond.get("c_crossattn")
            ct = cond.get("c_concat")
            cc = torch.cat(cc, 1) if isinstance(cc, (list, tuple)) else cc
            ct = torch.cat(ct, 1) if isinstance(ct, (list, tuple)) else ct
            return cc, ct
        if isinstance(cond, (list, tuple)):
            cond = torch.cat(list(cond), 1)
        if cond is not None and getattr(self.model.model, "conditioning_key", None) == "concat":
            return None, cond                    # DiffusionWrapper 'concat': the conditioning is the channel concat (ddpm.py:1407-1409)
        return cond, None

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1.,
               noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None,
               noise=None, use_graph=False, policy_batch=None, **kwargs):
        if conditioning is None and getattr(self.model.model, "conditioning_key", "crossattn") is not None:
            raise L.LdmkError("DDIMSampler.sample: conditioning is required (this model is cross-attention / concat "
                              "conditioned; the reference asserts the same in ddim2cond.py:80).  Only an unconditional LDM "
                              "(cond_stage_config '__is_unconditional__') samples with conditioning=None (ddim.py:77-84)")
        ctx, cat0 = self._split_cond(conditioning)
        first = ctx if ctx is not None else cat0
        if first is not None and first.shape[0] != batch_size:
            print(f"Warning: Got {first.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C_, H, W_ = shape
        size = (batch_size, C_, H, W_)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, noise=noise,
                                  use_graph=use_graph, policy_batch=policy_batch, mask=mask, x0=x0,
                                  quantize_denoised=quantize_x0, temperature=temperature, noise_dropout=noise_dropout,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                  mask_noise=kwargs.get("mask_noise"))

    @torch.no_grad()
    @rerun_if_flags_tripped(lambda self: self.model.model.diffusion_model)
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, noise=None,
                      use_graph=False, policy_batch=None, return_x_inter_only=False, invert=False, mask=None, x0=None,
                      quantize_denoised=False, temperature=1., noise_dropout=0., score_corrector=None,
                      corrector_kwargs=None, mask_noise=None, ddim_use_original_steps=False, timesteps=None, n_steps=None, **kwargs):
        """invert=True runs the forward DDIM (inversion) direction: index 0 -> S-1 with q_sample_ddim's update.
        ddim_use_original_steps / timesteps: ddim.py:127-134 -- every step of the model's own schedule (`timesteps`: only its first
        so many), or the first `int(min(timesteps / S, 1) S) - 1` entries of the DDIM subsequence.

        The reference's rarely used options (ddim.py:112-203; no shipped script sets them) run as device-side elementwise
        work around the same captured step, never on the host:
          mask / x0          inpainting blend before every step, `img = q_sample(x0, t) * mask + (1 - mask) * img` (:143-146);
                             `mask_noise` = optional per-step noise list for q_sample (parity with a seeded reference run)
          temperature        scales the step noise (:199)
          noise_dropout      F.dropout on the step noise (:200-201)
          quantize_denoised  pred_x0 snapped to the first stage's codebook before x_prev is formed (:195-196)
          score_corrector    `modify_score(model, e_t, x, t, c, **corrector_kwargs)` between the UNet and the update (:179-181);
                             a host callback, so it runs with eager launches."""
        dev = self.model.device
        unet = self.model.model.diffusion_model
        b = shape[0]
        S = self.ddim_timesteps.shape[0]
        img0 = torch.randn(shape, device=dev) if x_T is None else x_T.to(dev, torch.float32)
        ctx, cat = self._split_cond(cond)
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        nb = 2 * b if cfg else b
        y_in = None
        if self._is_adm():
            # class-conditional model (ddpm.py:1417-1419): the conditioning is the label vector y, which reaches the UNet as rows of
            # its label embedding added to the timestep embedding -- constant over the run, written once into the program's
            # `y_emb` input; guidance doubles it as [uncond | cond] like any conditioning (ddim.py:175)
            y_in = ctx if not cfg else torch.cat([self._split_cond(unconditional_conditioning)[0], ctx])
            assert y_in is not None and y_in.dim() == 1 and cat is None, "conditioning_key 'adm': conditioning = class labels (B,)"
            ctx_in, cat_in = None, None
        elif cfg:
            uc, ucat = self._split_cond(unconditional_conditioning)
            if ctx is None:                                # 'concat' conditioning: the guidance halves differ in the concat tensor
                ctx_in, cat_in = None, torch.cat([ucat, cat])
            else:
                ctx_in = torch.cat([uc, ctx])              # ddim.py:175: [uncond | cond]
                cat_in = None if cat is None else torch.cat([cat] * 2)
        else:
            ctx_in, cat_in = ctx, cat
        ncat = 0 if cat_in is None else cat_in.shape[1]
        L_ctx = 0 if ctx_in is None else ctx_in.shape[1]          # 0: unconditional UNet (no cross-attention)
        unet.policy_batch = None if policy_batch is None else (2 * policy_batch if cfg else policy_batch)
        pg = unet.program(nb, shape[2], shape[3], L_ctx, ncat)
        lib = pg.lib
        x_buf = pg.inputs["x"]
        if L_ctx:
            pg.inputs["context"].copy_(ctx_in.reshape(nb * L_ctx, -1))
        if ncat:
            pg.inputs["c_concat"].copy_(cat_in)
        if y_in is not None:
            pg.inputs["y_emb"].copy_(unet.label_emb.weight.detach().float()[y_in.to(dev, torch.int64)])
        pg.ctx_program.run()                               # context-only projections: once per sample() call
        img = x_buf[:b]                                    # the latent lives in the UNet's input buffer
        img.copy_(img0)
        if cfg:
            x_buf[b:].copy_(img0)
        need_noise = self._eta != 0. or noise is not None
        if mask is not None:
            assert x0 is not None, "mask needs x0 (ddim.py:144)"
            mask = mask.to(dev, torch.float32).expand(shape).contiguous()
            x0 = x0.to(dev, torch.float32)
        if score_corrector is not None:
            assert self.model.parameterization == "eps"
            use_graph = False                              # a host callback sits inside the step
        extras = mask is not None or quantize_denoised or score_corrector is not None or noise_dropout > 0. or temperature != 1.
        table, ts_table, tsteps = (self._inv_table if invert else self._table), self._ts_table, self.ddim_timesteps
        n_run = S
        if ddim_use_original_steps:
            assert not invert, "the inversion runs on the DDIM subsequence"
            table, ts_table, tsteps = self._original_tables()
            S = tsteps.shape[0]
            n_run = S if timesteps is None else int(timesteps)
        elif timesteps is not None:
            assert not invert
            n_run = int(min(timesteps / S, 1) * S) - 1       # ddim.py:129-131
        if n_steps is not None:                              # decode(): exactly the first n_steps entries of the schedule in use
            assert not invert and timesteps is None
            n_run = int(n_steps)
        assert 0 < n_run <= S, (n_run, S)
        # loop state (buffers, the captured step) lives ON the launch program it was captured over and dies with it: when the model
        # drops its programs (a re-pack, an arithmetic fall-back) nothing keeps the old workspace or its hipGraph alive, and a
        # new program can never be handed buffers of another batch size (rounds 2-4 keyed a sampler-side cache by id(pg))
        loops = pg.__dict__.setdefault("_ddim_loops", {})
        lkey = (cfg, float(unconditional_guidance_scale), self._sched_key, need_noise, bool(use_graph), noise is None, bool(invert),
                bool(ddim_use_original_steps), n_run)
        st = None if extras else loops.get(lkey)           # (option runs capture their own tensors: never cached)
        if st is None:
            st = dict(pred_x0=torch.empty_like(img0), step_idx=torch.zeros(1, dtype=torch.int32, device=dev),
                      nz=torch.empty_like(img0) if need_noise else None, graph=None)
            if not extras:
                loops[lkey] = st
        pred_x0, step_idx, nz_buf = st["pred_x0"], st["step_idx"], st["nz"]
        eps = pg.outputs["eps"]
        per = img0[0].numel()
        scale = float(unconditional_guidance_scale)
        first = 0 if invert else n_run - 1
        adv = -1 if invert else 1

        def reset_state():
            img.copy_(img0)
            if cfg:
                x_buf[b:].copy_(img0)
            step_idx.fill_(first)
            pg.inputs["t"].fill_(int(tsteps[first]))

        mnz = torch.zeros_like(img0) if mask is not None else None      # (zeros: the graph warm-up steps run before the first fill)
        sqrt_ac, sqrt_1mac = self.model.sqrt_alphas_cumprod, self.model.sqrt_one_minus_alphas_cumprod
        eps_c = torch.empty_like(img0) if score_corrector is not None else None

        def one_step():
            if mask is not None:                           # ddim.py:143-146, timestep read from the device-side vector
                tcur = pg.inputs["t"][:b]
                img_orig = (sqrt_ac.gather(-1, tcur).view(b, 1, 1, 1) * x0 + sqrt_1mac.gather(-1, tcur).view(b, 1, 1, 1) * mnz)
                img.copy_(img_orig * mask + (1. - mask) * img)
                if cfg:
                    x_buf[b:].copy_(img)
            if nz_buf is not None and (temperature != 1. or noise_dropout > 0.):
                if temperature != 1.:
                    nz_buf.mul_(temperature)
                if noise_dropout > 0.:
                    nz_buf.copy_(torch.nn.functional.dropout(nz_buf, p=noise_dropout))
            a_prev = table[step_idx.long(), 1] if quantize_denoised else None      # read before the kernel advances the counter
            pg.run()
            e_ptr, k_cfg = eps.data_ptr(), (1 if cfg else 0)
            if score_corrector is not None:                # ddim.py:179-181 acts on the guidance-combined score
                e_t = eps[:b] if not cfg else eps[:b] + scale * (eps[b:] - eps[:b])
                tcur = pg.inputs["t"][:b].clone()
                eps_c.copy_(score_corrector.modify_score(self.model, e_t, img.clone(), tcur, cond, **(corrector_kwargs or {})))
                e_ptr, k_cfg = eps_c.data_ptr(), 0
            rc = lib.ldmk_ddim_step(img.data_ptr(), e_ptr, 0 if nz_buf is None else nz_buf.data_ptr(),
                                    table.data_ptr(), step_idx.data_ptr(), scale, k_cfg, img.data_ptr(),
                                    pred_x0.data_ptr(), per, b, ts_table.data_ptr(), pg.inputs["t"].data_ptr(), nb, adv, S,
                                    torch.cuda.current_stream().cuda_stream)
            L.check(rc, "ldmk_ddim_step")
            if quantize_denoised:                          # x_prev = sqrt(a_prev) Q(pred_x0) + dir_xt + noise (ddim.py:194-202)
                q = self.model.first_stage_model.quantize(pred_x0)[0]
                img.add_(torch.sqrt(a_prev) * (q - pred_x0))
                pred_x0.copy_(q)
            if cfg:
                x_buf[b:].copy_(img)                       # both CFG halves see the same latent (ddim.py:173)

        reset_state()
        step = one_step
        if use_graph:
            if st["graph"] is None:
                if (nz_buf is not None and noise is None) or (mask is not None and mask_noise is None):
                    def step_with_noise():
                        if nz_buf is not None and noise is None:
                            nz_buf.normal_()
                        if mask is not None and mask_noise is None:
                            mnz.normal_()
                        one_step()
                    st["graph"] = GraphedProgram(step_with_noise)
                else:
                    st["graph"] = GraphedProgram(one_step)
                reset_state()                              # warm-up + capture advanced the device state
            step = st["graph"].replay

        intermediates = {"x_inter": [img0], "pred_x0": [img0]}
        for i in range(n_run):
            index = n_run - i - 1
            if nz_buf is not None:
                if noise is not None:
                    nz_buf.copy_(noise[i])
                elif not use_graph:
                    nz_buf.normal_()
            if mask is not None:
                if mask_noise is not None:
                    mnz.copy_(mask_noise[i])
                elif not use_graph:
                    mnz.normal_()
            step()
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == n_run - 1:
                intermediates["x_inter"].append(img.clone())
                intermediates["pred_x0"].append(pred_x0.clone())
        out = img.clone()
        if return_x_inter_only:                            # TF sampler returns the x_inter list, ddim2cond.py:156
            return out, intermediates["x_inter"]
        return out, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, noise=None):
        """Single step with the reference's signature (ddim.py:164-203); returns (x_prev, pred_x0).  use_original_steps: `index`
        counts the model's own timesteps (`_original_tables`)."""
        dev = x.device
        b = x.shape[0]
        table = self._original_tables()[0] if use_original_steps else self._table
        ctx, cat = self._split_cond(c)
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        if cfg:
            uc, ucat = self._split_cond(unconditional_conditioning)
            x2, t2 = torch.cat([x] * 2), torch.cat([t] * 2)
            if ctx is None:       # 'concat' conditioning: the guidance halves differ in the concat tensor ([uncond | cond], as in ddim_sampling)
                e = self.model.apply_model(x2, t2, None, torch.cat([ucat, cat]))
            else:
                e = self.model.apply_model(x2, t2, torch.cat([uc, ctx]), None if cat is None else torch.cat([cat] * 2))
        else:
            e = self.model.apply_model(x, t, ctx, cat)
        if score_corrector is not None:                      # ddim.py:179-181, on the guidance-combined score
            assert self.model.parameterization == "eps"
            if cfg:
                e_u, e_c = e.chunk(2)
                e = e_u + unconditional_guidance_scale * (e_c - e_u)
                cfg = False
            e = score_corrector.modify_score(self.model, e, x, t, c, **(corrector_kwargs or {})).contiguous()
        if self._eta != 0.:
            noise = (torch.randn_like(x) if noise is None else noise.to(dev)) * temperature
            if noise_dropout > 0.:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        step_idx = torch.full((1,), int(index), dtype=torch.int32, device=dev)
        x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
        L.call("ldmk_ddim_step", x.contiguous().data_ptr(), e.data_ptr(), 0 if noise is None else noise.contiguous().data_ptr(),
               table.data_ptr(), step_idx.data_ptr(), float(unconditional_guidance_scale), 1 if cfg else 0,
               x_prev.data_ptr(), pred_x0.data_ptr(), x[0].numel(), b, 0, 0, 0, 0, 0,
               torch.cuda.current_stream().cuda_stream)
        if quantize_denoised:                                # ddim.py:195-196
            q = self.model.first_stage_model.quantize(pred_x0)[0]
            x_prev = x_prev + torch.sqrt(table[int(index), 1]) * (q - pred_x0)
            pred_x0 = q
        return x_prev, pred_x0

    # ------------------------------------------------------------------------------------------ img2img (SDEdit) pair
    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """ddim.py:205-219 / ddim2cond.py:198-212: x_t = sqrt(a[t]) x0 + sqrt(1 - a[t]) noise with `t` indexing the DDIM subsequence
        of the last make_schedule -- or, use_original_steps, the model's own timesteps.  One ldmk_q_sample launch.  The tables are
        the SAMPLER's (float32 roots of the float32 alphas, make_schedule :35-36,47-50), not the model's float64-rounded buffers."""
        dev = self.model.device
        x0 = x0.to(dev, torch.float32).contiguous()
        if use_original_steps:
            ac = self.model.alphas_cumprod.to(dev, torch.float32)
            a, b = torch.sqrt(ac), torch.sqrt(1. - ac)
        else:
            a = torch.sqrt(self.ddim_alphas.to(dev, torch.float32)).contiguous()
            b = torch.as_tensor(np.asarray(self.ddim_sqrt_one_minus_alphas), dtype=torch.float32).to(dev).contiguous()
        t = t.to(dev, torch.int64).contiguous()
        assert t.shape == (x0.shape[0],) and int(t.max()) < a.shape[0] and int(t.min()) >= 0, "stochastic_encode: t out of the table"
        noise = torch.randn_like(x0) if noise is None else noise.to(dev, torch.float32).contiguous()
        out = torch.empty_like(x0)
        L.call("ldmk_q_sample", x0.data_ptr(), noise.data_ptr(), t.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(),
               x0.shape[0], x0[0].numel(), torch.cuda.current_stream().cuda_stream)
        return out

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, use_graph=False, noise=None):
        """ddim2cond.py:230-250: the DDIM updates of the first `t_start` entries of the schedule in use (the subsequence of the last
        make_schedule, or the model's own timesteps), from index t_start - 1 down to 0 -- the device-resident loop of
        ddim_sampling started there.  `noise`: optional per-step list for eta > 0 (parity with a seeded reference run)."""
        out, _ = self.ddim_sampling(cond, tuple(x_latent.shape), x_T=x_latent, unconditional_guidance_scale=unconditional_guidance_scale,
                                    unconditional_conditioning=unconditional_conditioning,
                                    ddim_use_original_steps=use_original_steps, n_steps=int(t_start), use_graph=use_graph, noise=noise)
        return out

    # ------------------------------------------------------------------------------------------ latent manipulation
    @torch.no_grad()
    def compute_latents(self, S, batch_size, shape, conditioning=None, x0=None, eta=0., unconditional_guidance_scale=1.,
                        unconditional_conditioning=None, strength=0.5, verbose=True, use_graph=False, **kwargs):
        """DDIM inversion followed by regeneration (face_reenactment/compute_latents.py:297-362):
        returns (img, x_latent, x0).  The forward direction is `q_sample_ddim` (:364-406)."""
        assert x0 is not None and conditioning is not None
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose, strength=strength)
        size = (batch_size,) + tuple(shape)
        kw = dict(unconditional_guidance_scale=unconditional_guidance_scale,
                  unconditional_conditioning=unconditional_conditioning, use_graph=use_graph)
        x_lat, _ = self.ddim_sampling(conditioning, size, x_T=x0, invert=True, **kw)
        img, _ = self.ddim_sampling(conditioning, size, x_T=x_lat, **kw)
        return img, x_lat, x0

    @torch.no_grad()
    def latent_manipulation(self, c_src, c_trg, S, batch_size, shape, x0=None, eta=0., unconditional_guidance_scale=1.,
                            unconditional_conditioning=None, strength=1.0, x_T=None, verbose=True, use_graph=False, **kwargs):
        """The emotion edit of face_reenactment/latent_manipulation.py:420-490: DDIM inversion of x0 under the SOURCE conditioning
        (`q_sample_ddim`, :377-418), regeneration under the TARGET conditioning, both on the strength-scaled schedule and on the
        device-resident loop (the two directions are two captured steps of the same launch program).  Returns (img, x_latent, x0)."""
        assert c_src is not None and c_trg is not None
        assert x0 is not None
        assert x_T is None
        assert eta == 0
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose, strength=strength)
        size = (batch_size,) + tuple(shape)
        kw = dict(unconditional_guidance_scale=unconditional_guidance_scale,
                  unconditional_conditioning=unconditional_conditioning, use_graph=use_graph)
        x_lat, _ = self.ddim_sampling(c_src, size, x_T=x0, invert=True, **kw)
        img, _ = self.ddim_sampling(c_trg, size, x_T=x_lat, **kw)
        return img, x_lat, x0

    @torch.no_grad()
    def ddim_tuned_sampling(self, S, batch_size, shape, x_lat, cond, eta=0., unconditional_guidance_scale=1.,
                            unconditional_conditioning=None, strength=1.0, verbose=True, use_graph=False, **kwargs):
        """Reverse DDIM from a precomputed latent on the strength-scaled schedule
        (face_reenactment/latent_manipulation_tuned.py:493-538)."""
        assert cond is not None and x_lat is not None and eta == 0
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose, strength=strength)
        img, _ = self.ddim_sampling(cond, (batch_size,) + tuple(shape), x_T=x_lat,
                                    unconditional_guidance_scale=unconditional_guidance_scale,
                                    unconditional_conditioning=unconditional_conditioning, use_graph=use_graph)
        return img

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def progressive_sampling(self, c1, xid, xmasks, audio_feats, S, batch_size, num_frames, shape, audio_window,
                             eta=0., verbose=True, unconditional_guidance_scale=1., unconditional_conditioning=None,
                             x_T=None, fixed_identity=False, use_graph=True, policy_batch=None, clips=None, **kwargs):
        """Talking-face clip generation, progressive_sampling_difftalk.py:245-319.

        clips=V: V independent videos advanced in lock step (see `_progressive_clips`; the reference's run is a loop over 150
          test videos, progressive_sampling_difftalk.py:336): c1 (V,1,D), xid (V,c,h,w), xmasks / audio_feats / x_T lists of V
          per-clip tensors; returns a list of V frame lists.

        fixed_identity=False: the reference's behaviour -- frames are a serial chain, the identity latent of
          frame k+1 is the latent generated for frame k (:316-317); batch 1, one captured step replayed S times
          per frame.
        fixed_identity=True: identity latent = encode(identity image) for every frame, which makes frames
          independent: the whole clip is ONE batched DDIMSampler.sample call (and shards across GPUs).
        x_T: optional (T,1,C,H,W) start noise per frame (the reference draws torch.randn per frame).
        Returns (list of T latents (1,C,H,W), None) like the reference.
        """
        assert c1 is not None
        assert eta in [0., 1.]
        if unconditional_guidance_scale != 1. and unconditional_conditioning is not None:
            raise NotImplementedError("progressive_sampling: the reference's CFG branch raises (torch.cat(..., dim=21), "
                                      "progressive_sampling_difftalk.py:299); only scale=1 is defined")
        if clips is not None:
            assert not fixed_identity, "fixed-identity frames are independent already: one batched sample() call"
            return self._progressive_clips(int(clips), c1, xid, xmasks, audio_feats, S, shape, audio_window, eta, x_T,
                                           use_graph, policy_batch), None
        m = self.model
        if audio_feats.dim() == 3:
            assert audio_feats.shape[0] == 1
            audio_feats = audio_feats.squeeze(0)
        T = audio_feats.shape[0]
        C_, H, W_ = shape
        dev = m.device
        idx = torch.tensor([[min(max(f + i, 0), T - 1) for i in range(-audio_window, audio_window + 1)]
                            for f in range(T)], device=audio_feats.device)
        c2_all = m.cond_stage_model_2(audio_feats[idx])                       # (T,1,768): one batched call
        c12_all = torch.cat([c1.expand(T, -1, -1), c2_all], dim=2)            # (T,1,1024)
        c3_all = m.encode_first_stage(xmasks)                                  # (T,c,h,w): one batched encode
        if x_T is None:
            x_T = torch.randn(T, batch_size, C_, H, W_, device=dev)
        if fixed_identity:
            c34 = torch.cat([c3_all, xid.expand(T, -1, -1, -1)], dim=1)
            out, _ = self.sample(S, T, shape, {C12: c12_all, C34: c34}, eta=eta, x_T=x_T[:, 0], verbose=False,
                                 use_graph=use_graph, policy_batch=policy_batch)
            return [out[f:f + 1] for f in range(T)], None
        zid = xid.clone()
        frames = []
        for f in range(T):
            c = {C12: c12_all[f:f + 1], C34: torch.cat([c3_all[f:f + 1], zid], dim=1)}
            img, _ = self.sample(S, batch_size, shape, c, eta=eta, x_T=x_T[f], verbose=False, use_graph=use_graph,
                                 policy_batch=policy_batch)
            frames.append(img)
            zid = img.clone()
        return frames, None

    @torch.no_grad()
    def _progressive_clips(self, V, c1, xid, xmasks, audio_feats, S, shape, audio_window, eta, x_T, use_graph, policy_batch):
        """V independent talking-face videos, each the reference's serial chain (frame k+1's identity latent = frame k's
        result, progressive_sampling_difftalk.py:282-317), advanced frame by frame TOGETHER: frame f of every clip that still has
        one is ONE batch of the batched program, so the autoregressive mode runs at the batched rate instead of batch 1.  Clips
        may have different lengths: a finished clip leaves the batch.  Tile plans are made for `policy_batch` (default V) whatever
        the active count, so every clip's frames are bit for bit those of `progressive_sampling` on that clip alone with the same
        `policy_batch` (a sample's result does not depend on what else is in its batch).  Across GPUs: replicas only -- different
        clips per rank, no collective."""
        m = self.model
        assert len(xmasks) == V and len(audio_feats) == V and c1.shape[0] == V and xid.shape[0] == V
        C_, H, W_ = shape
        dev = m.device
        pol = V if policy_batch is None else policy_batch
        c12, c3, Ts = [], [], []
        for v in range(V):                                   # per-clip conditioning, exactly the single-clip calls
            af = audio_feats[v]
            if af.dim() == 3:
                assert af.shape[0] == 1
                af = af.squeeze(0)
            T = af.shape[0]
            idx = torch.tensor([[min(max(f + i, 0), T - 1) for i in range(-audio_window, audio_window + 1)]
                                for f in range(T)], device=af.device)
            c2_all = m.cond_stage_model_2(af[idx])
            c12.append(torch.cat([c1[v:v + 1].expand(T, -1, -1), c2_all], dim=2))
            c3.append(m.encode_first_stage(xmasks[v]))
            Ts.append(T)
        if x_T is None:
            x_T = [torch.randn(T, 1, C_, H, W_, device=dev) for T in Ts]
        zid = [xid[v:v + 1].clone() for v in range(V)]
        frames = [[] for _ in range(V)]
        for f in range(max(Ts)):
            act = [v for v in range(V) if f < Ts[v]]
            c = {C12: torch.cat([c12[v][f:f + 1] for v in act]),
                 C34: torch.cat([torch.cat([c3[v][f:f + 1], zid[v]], dim=1) for v in act])}
            img, _ = self.sample(S, len(act), shape, c, eta=eta, x_T=torch.cat([x_T[v][f] for v in act]), verbose=False,
                                 use_graph=use_graph, policy_batch=pol)
            for i, v in enumerate(act):
                frames[v].append(img[i:i + 1].clone())
                zid[v] = img[i:i + 1].clone()
        return frames
