"""Differentiable first stage and differentiable DDIM (SURVEY §8f "next" row N2).

Reference: `LatentDiffusionCLIP.forward` (latent_diffclip.py:969-1003): a few DDIM steps with gradients
(`differentiable_p_sample_ddim`, ddim2.py:252-290, CFG by batch doubling), then `differentiable_decode_first_stage`
(ddpm.py:767-824 = VQModelInterface.decode without no_grad: quantize with the straight-through estimator,
post_quant_conv, Decoder), then image-space losses (l2 / ArcFace / CLIP) that stay on PyTorch-ROCm.  The first stage is
frozen, so the decoder needs data gradients only; the UNet accumulates parameter gradients over the steps.

`DecoderGrad`   : VQGAN decode with a tape; backward(d image) -> d latent.  Kernels: ldmk_igemm (mirrored-tap weights,
                  b_trans), ldmk_gn_bwd, batched GEMMs + ldmk_softmax_bwd_rows for the single-head AttnBlock.
`DifferentiableDDIM`: S steps of x_{i-1} = c1*x_i + c2*eps_theta(x_i) (eta = 0) through `UNetTrainer` passes.
"""
import torch
import torch.nn.functional as F

from . import lib as L
from . import ops
from . import train_ops as T
from .train import UNetTrainer, gemm


class DecoderGrad:
    def __init__(self, vq):
        vq._ensure()
        self.vq, self.P, self.sd = vq, vq._packed, vq._sd
        self.dev = next(vq.parameters()).device
        self._wd = {}          # cached data-gradient weights (the first stage is frozen)
        d = vq.decoder
        self.wpad_in = ops.pack_conv3x3(F.pad(self.sd["decoder.conv_in.weight"], (0, 0, 0, 0, 0, 32 - d.z_channels)).contiguous())
        self.wpad_out = ops.pack_conv3x3(F.pad(self.sd["decoder.conv_out.weight"], (0, 0, 0, 0, 0, 0, 0, 32 - d.out_ch)).contiguous())
        self.bpad_out = F.pad(self.sd["decoder.conv_out.bias"], (0, 32 - d.out_ch))
        self.tape = []

    # ---- primitives ------------------------------------------------------------------------------------------
    def _conv(self, x4, wp, bias, upsample=False, residual=None):
        n, h, w_, c = x4.shape
        cout = wp.shape[1]
        oh, ow = (2 * h, 2 * w_) if upsample else (h, w_)
        out = torch.empty(n, oh, ow, cout, device=self.dev)
        a = ops.make_igemm_args(n * oh * ow, cout, 9 * c, x4, c, wp, out, cout, oh * ow,
                                conv=(h, w_, oh, ow, 1, 1, 1 if upsample else 0), bias=bias, residual=residual)
        gemm(a, self.dev)
        return out

    def _conv_dx(self, dy4, key, wp, cin, in_hw, upsample=False):
        wd = self._wd.get(key)
        if wd is None:
            wd = self._wd[key] = T.pack_dgrad3x3(wp, cin, wp.shape[1])
        if upsample:
            return T.sumpool2(T.conv3x3_dgrad(dy4, wd, (2 * in_hw[0], 2 * in_hw[1])))
        return T.conv3x3_dgrad(dy4, wd, in_hw)

    def _lin(self, x2d, wp, bias, rows_per_sample, residual=None):
        M, N = x2d.shape[0], wp.shape[1]
        out = torch.empty(M, N, device=self.dev)
        a = ops.make_igemm_args(M, N, x2d.shape[1], x2d, x2d.shape[1], wp, out, N, rows_per_sample, bias=bias, residual=residual)
        gemm(a, self.dev)
        return out

    def _lin_dx(self, dy, wp, out=None, acc=False):
        M, N = dy.shape
        K = wp.shape[0]
        out = torch.empty(M, K, device=self.dev) if out is None else out
        a = ops.make_igemm_args(M, K, N, dy, N, wp, out, K, M, b_trans=True, ldb=wp.stride(0), residual=out if acc else None)
        gemm(a, self.dev)
        return out

    def _gn(self, x4, prefix, silu):
        n, h, w_, c = x4.shape
        hw = h * w_
        chunks = L.load().ldmk_gn_chunks(hw)
        partial = torch.empty(n * chunks * c * 3, device=self.dev)
        coef = torch.empty(n, 2, c, device=self.dev)
        ops.gn_coef(x4, None, n, hw, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"], 1e-6, partial=partial, coef=coef)
        mr = T.gn_group_stats(partial, c, None, 0, n, hw, 32, 1e-6)
        return ops.gn_apply(x4, None, coef, n, hw, silu=silu), (coef, mr)

    def _gn_dx(self, dy2d, x4, prefix, saved, silu, out=None, acc=False):
        coef, mr = saved
        n, h, w_, c = x4.shape
        out = torch.empty_like(x4) if out is None else out
        dump = torch.empty(2 * c, device=self.dev)            # frozen norm parameters: their gradients are discarded
        T.gn_bwd(x4, None, dy2d, coef, mr, self.sd[prefix + ".weight"], n, h * w_, silu=silu, dx0=out, acc0=acc,
                 dgamma=dump[:c], dbeta=dump[c:])
        return out

    # ---- blocks (model.py:95-129 ResnetBlock, :157-202 AttnBlock) ----------------------------------------------
    def _resnet(self, prefix, m, x):
        n, h, w_, _ = x.shape
        P, sd = self.P, self.sd
        y1, s1 = self._gn(x, prefix + "norm1", True)
        h1 = self._conv(y1.view(n, h, w_, m.cin), P[prefix + "conv1.weight"], sd[prefix + "conv1.bias"])
        y2, s2 = self._gn(h1, prefix + "norm2", True)
        if m.cin != m.cout:
            sk = self._lin(x.reshape(n * h * w_, m.cin), P[prefix + "nin_shortcut.weight"], sd[prefix + "nin_shortcut.bias"], h * w_)
            out = self._conv(y2.view(n, h, w_, m.cout), P[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], residual=sk)
        else:
            out = self._conv(y2.view(n, h, w_, m.cout), P[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], residual=x)

        def bwd(dout):
            dy2 = self._conv_dx(dout, prefix + "conv2", P[prefix + "conv2.weight"], m.cout, (h, w_))
            dh1 = self._gn_dx(dy2.view(n * h * w_, -1), h1, prefix + "norm2", s2, True)
            dy1 = self._conv_dx(dh1, prefix + "conv1", P[prefix + "conv1.weight"], m.cin, (h, w_))
            dx = self._gn_dx(dy1.view(n * h * w_, -1), x, prefix + "norm1", s1, True)
            if m.cin != m.cout:
                self._lin_dx(dout.view(n * h * w_, -1), P[prefix + "nin_shortcut.weight"], out=dx.view(n * h * w_, -1), acc=True)
            else:
                T.axpy_(dx, dout, 1.0)
            return dx
        self.tape.append(bwd)
        return out

    def _attn(self, prefix, m, x):
        n, h, w_, c = x.shape
        hw, rows = h * w_, n * h * w_
        P, sd = self.P, self.sd
        scale = float(int(c) ** (-0.5))
        xn, sx = self._gn(x, prefix + "norm", False)
        q = self._lin(xn, P[prefix + "q.weight"], sd[prefix + "q.bias"], hw)
        k = self._lin(xn, P[prefix + "k.weight"], sd[prefix + "k.bias"], hw)
        v = self._lin(xn, P[prefix + "v.weight"], sd[prefix + "v.bias"], hw)
        p = ops.bmm(q.view(n, hw, c), k.view(n, hw, c), True)
        ops.softmax_rows_(p.view(rows, hw), scale)
        o = ops.bmm(p, v.view(n, hw, c), False).view(rows, c)
        out = self._lin(o, P[prefix + "proj_out.weight"], sd[prefix + "proj_out.bias"], hw, residual=x.view(rows, c)).view(n, h, w_, c)

        def bwd(dout):
            d2 = dout.view(rows, c)
            do = self._lin_dx(d2, P[prefix + "proj_out.weight"]).view(n, hw, c)
            dv = torch.empty(n, hw, c, device=self.dev)
            T._wgrad_batched(p, do, dv, n, hw, hw, c)                       # dV = P^T dO
            dp = ops.bmm(do, v.view(n, hw, c), True)                        # dP = dO V^T
            T.softmax_bwd_rows_(p.view(rows, hw), dp.view(rows, hw), scale)
            dq = ops.bmm(dp, k.view(n, hw, c), False)                       # dQ = dS K
            dk = torch.empty(n, hw, c, device=self.dev)
            T._wgrad_batched(dp, q.view(n, hw, c), dk, n, hw, hw, c)        # dK = dS^T Q
            dxn = self._lin_dx(dq.view(rows, c), P[prefix + "q.weight"])
            self._lin_dx(dk.view(rows, c), P[prefix + "k.weight"], out=dxn, acc=True)
            self._lin_dx(dv.view(rows, c), P[prefix + "v.weight"], out=dxn, acc=True)
            dx = self._gn_dx(dxn, x, prefix + "norm", sx, False)
            T.axpy_(dx, dout, 1.0)
            return dx
        self.tape.append(bwd)
        return out

    # ---- decode ------------------------------------------------------------------------------------------
    def forward(self, z, force_not_quantize=False):
        """z (n,zc,h,w) NCHW -> image (n,out_ch,H,W); autoencoder.py:274-282 + model.py:535-568, with a tape."""
        vq, d, P, sd = self.vq, self.vq.decoder, self.P, self.sd
        if not z.is_cuda:
            raise L.LdmkError("DecoderGrad.forward: CUDA tensors only (no CPU fallback)")
        n, zc, h, w_ = z.shape
        self.tape = []
        zq = z.contiguous().float()
        if not force_not_quantize:
            zq, _ = ops.vq_nearest(zq, sd["quantize.embedding.weight"])     # straight-through: d zq / d z = identity
        pq = ops.conv1x1_nchw(zq, P["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        xp = torch.zeros(n, h, w_, 32, device=self.dev)
        xp[..., :zc] = pq.permute(0, 2, 3, 1)
        top = d.ch * d.ch_mult[-1]
        x = self._conv(xp, self.wpad_in, sd["decoder.conv_in.bias"])

        def bwd_in(dx):
            dxp = self._conv_dx(dx, "conv_in", self.wpad_in, 32, (h, w_))
            dpq = dxp[..., :zc].permute(0, 3, 1, 2).contiguous()
            wt = P["post_quant_conv.weight"].t().contiguous()              # [cin][cout]: the transposed 1x1
            return ops.conv1x1_nchw(dpq, wt, torch.zeros(zc, device=self.dev))
        self.tape.append(bwd_in)

        x = self._resnet("decoder.mid.block_1.", d.mid.block_1, x)
        x = self._attn("decoder.mid.attn_1.", d.mid.attn_1, x)
        x = self._resnet("decoder.mid.block_2.", d.mid.block_2, x)
        for lvl in reversed(range(d.num_resolutions)):
            up = d.up[lvl]
            for ib in range(d.num_res_blocks + 1):
                x = self._resnet(f"decoder.up.{lvl}.block.{ib}.", up.block[ib], x)
                if len(up.attn) > 0:
                    x = self._attn(f"decoder.up.{lvl}.attn.{ib}.", up.attn[ib], x)
            if lvl != 0:
                key = f"decoder.up.{lvl}.upsample.conv"
                xin = x
                x = self._conv(xin, P[key + ".weight"], sd[key + ".bias"], upsample=True)
                self.tape.append(lambda dy, key=key, xin=xin: self._conv_dx(dy, key, P[key + ".weight"], xin.shape[-1],
                                                                             (xin.shape[1], xin.shape[2]), upsample=True))
        xf = x
        nf, hf, wf, cf = xf.shape
        y, sy = self._gn(xf, "decoder.norm_out", True)
        img_pad = self._conv(y.view(nf, hf, wf, cf), self.wpad_out, self.bpad_out)

        def bwd_out(dimg_pad):
            dy = self._conv_dx(dimg_pad, "conv_out", self.wpad_out, cf, (hf, wf))
            return self._gn_dx(dy.view(nf * hf * wf, -1), xf, "decoder.norm_out", sy, True)
        self.tape.append(bwd_out)
        return img_pad[..., :d.out_ch].permute(0, 3, 1, 2).contiguous()

    def backward(self, dimg):
        """dimg (n,out_ch,H,W) -> d(loss)/d(z) (n,zc,h,w)."""
        g = UNetTrainer.pad_output_grad(dimg.float())
        for fn in reversed(self.tape):
            g = fn(g)
        self.tape = []
        return g


def lincomb(terms):
    """sum_i a_i * x_i on the device (ldmk_axpy), for the few latent-sized linear combinations of the DDIM update."""
    out = torch.zeros_like(terms[0][1], memory_format=torch.contiguous_format)
    for a, x in terms:
        T.axpy_(out, x.contiguous(), float(a))
    return out


class DifferentiableDDIM:
    """`LatentDiffusionCLIP.forward` up to the image (latent_diffclip.py:969-1003) with gradients: eta = 0 DDIM steps
    (ddim2.py:252-290) with classifier-free guidance by batch doubling, then the differentiable decode."""

    def __init__(self, model, trainer=None, decoder=None):
        self.model = model
        self.tr = trainer if trainer is not None else model.trainer()
        self.dec = decoder if decoder is not None else DecoderGrad(model.first_stage_model)

    def forward(self, x, c, table, timesteps, scale=1.0, uc=None):
        """x (n,C,H,W) start latent; c / uc (n,1,ctx) condition / null tokens; `table` [S][4] = (a_t, a_prev, sigma,
        sqrt(1-a_t)) rows as built by schedule.ddim_step_table (sigma must be 0); timesteps (S,) int.  Walks the table
        from the last row to the first (np.flip(ddim_timesteps)) and returns the decoded image."""
        self.passes = []
        n = x.shape[0]
        cfg = uc is not None and scale != 1.0
        tr = self.tr
        x = x.float()
        for i in reversed(range(len(timesteps))):
            a_t, a_prev, sigma, s1m = (float(v) for v in table[i])
            assert sigma == 0.0, "differentiable DDIM is deterministic (eta = 0) in the reference's fine-tuning scripts"
            ts = torch.full((n,), int(timesteps[i]), device=x.device, dtype=torch.long)
            if cfg:
                eps2 = tr.forward(torch.cat([x, x]), torch.cat([ts, ts]), torch.cat([uc, c]))
                e_t = lincomb([(1.0 - scale, eps2[:n]), (scale, eps2[n:])])      # e_u + s*(e_c - e_u)
            else:
                e_t = tr.forward(x, ts, c)
            # x_prev = sqrt(a_prev) * (x - s1m*e)/sqrt(a_t) + sqrt(1 - a_prev) * e   (linear in x and e)
            cx = (a_prev / a_t) ** 0.5
            ce = (1.0 - a_prev) ** 0.5 - cx * s1m
            self.passes.append((tr.last_pass, cx, ce, cfg, scale, n))
            x = lincomb([(cx, x), (ce, e_t)])
        self.z = x
        return self.dec.forward(lincomb([(1.0 / float(self.model.scale_factor), x)]))

    def backward(self, dimg):
        """Accumulates the UNet parameter gradients of all steps into trainer.P.grad; returns d(loss)/d(x_start)."""
        tr = self.tr
        dx = lincomb([(1.0 / float(self.model.scale_factor), self.dec.backward(dimg))])
        tr.P.grad.zero_()
        old = (tr.acc_params, tr.want_dx)
        tr.acc_params, tr.want_dx = True, True
        try:
            for ps, cx, ce, cfg, scale, n in reversed(self.passes):
                if cfg:
                    deps = torch.cat([lincomb([(ce * (1.0 - scale), dx)]), lincomb([(ce * scale, dx)])])
                    dxin = tr.backward(UNetTrainer.pad_output_grad(deps), ps)
                    dx = lincomb([(cx, dx), (1.0, dxin[:n]), (1.0, dxin[n:])])
                else:
                    dxin = tr.backward(UNetTrainer.pad_output_grad(lincomb([(ce, dx)])), ps)
                    dx = lincomb([(cx, dx), (1.0, dxin)])
        finally:
            tr.acc_params, tr.want_dx = old
        self.passes = []
        return dx
