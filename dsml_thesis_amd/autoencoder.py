"""MI355X-native VQGAN first stage: drop-in for `ldm.models.autoencoder.VQModelInterface`
(autoencoder.py:264-282) as the samplers use it -- `decode` (VQ lookup -> post_quant_conv ->
Decoder, model.py:462-568) and `encode` (Encoder -> quant_conv, model.py:368-459).  Training-side
members of the reference class (loss, optimisers, EMA of the VQGAN) are out of scope.

Parameter tree and state-dict keys equal the reference's (`encoder.*`, `decoder.*`,
`quantize.embedding.weight`, `quant_conv.*`, `post_quant_conv.*`).  Forward passes are launch
programs over libldmk.so kernels (NHWC inside, NCHW at the boundary); no PyTorch fallback.
"""
import math

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .engine import NetBuilder, Program
from .unet import _Params, _Slots, _conv_params, _norm_params


def _resnet(cin, cout):
    ch = dict(norm1=_norm_params(cin), conv1=_conv_params(cin, cout, 3), norm2=_norm_params(cout),
              conv2=_conv_params(cout, cout, 3))
    if cin != cout:
        ch["nin_shortcut"] = _conv_params(cin, cout, 1)
    m = _Slots(**ch)
    m.cin, m.cout = cin, cout
    return m


def _attn_block(c):
    m = _Slots(norm=_norm_params(c), q=_conv_params(c, c, 1), k=_conv_params(c, c, 1), v=_conv_params(c, c, 1),
               proj_out=_conv_params(c, c, 1))
    m.c = c
    return m


class _Level(nn.Module):
    pass


class Decoder(nn.Module):
    """Parameter tree of model.py:462-533."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if use_linear_attn or attn_type != "vanilla" or give_pre_end or tanh_out or not resamp_with_conv:
            raise NotImplementedError("Decoder: only the vanilla-attention / conv-resample configuration is built")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.z_channels, self.out_ch = resolution, z_channels, out_ch
        self.attn_resolutions = list(attn_resolutions)
        self.ch_mult = list(ch_mult)
        block_in = ch * ch_mult[-1]
        curr = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr, curr)
        self.conv_in = _conv_params(z_channels, block_in, 3)
        self.mid = _Slots(block_1=_resnet(block_in, block_in), attn_1=_attn_block(block_in),
                          block_2=_resnet(block_in, block_in))
        ups = []
        for lvl in reversed(range(self.num_resolutions)):
            block_out = ch * ch_mult[lvl]
            up = _Level()
            up.block = nn.ModuleList()
            up.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                up.block.append(_resnet(block_in, block_out))
                block_in = block_out
                if curr in self.attn_resolutions:
                    up.attn.append(_attn_block(block_in))
            if lvl != 0:
                up.upsample = _Slots(conv=_conv_params(block_in, block_in, 3))
                curr *= 2
            ups.insert(0, up)
        self.up = nn.ModuleList(ups)
        self.norm_out = _norm_params(block_in)
        self.conv_out = _conv_params(block_in, out_ch, 3)
        self._final_ch = block_in


class Encoder(nn.Module):
    """Parameter tree of model.py:368-432."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn or attn_type != "vanilla" or not resamp_with_conv:
            raise NotImplementedError("Encoder: only the vanilla-attention / conv-resample configuration is built")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.attn_resolutions = list(attn_resolutions)
        self.conv_in = _conv_params(in_channels, ch, 3)
        curr = resolution
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for lvl in range(self.num_resolutions):
            block_in = ch * in_mult[lvl]
            block_out = ch * ch_mult[lvl]
            down = _Level()
            down.block = nn.ModuleList()
            down.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                down.block.append(_resnet(block_in, block_out))
                block_in = block_out
                if curr in self.attn_resolutions:
                    down.attn.append(_attn_block(block_in))
            if lvl != self.num_resolutions - 1:
                down.downsample = _Slots(conv=_conv_params(block_in, block_in, 3))
                curr //= 2
            self.down.append(down)
        self.mid = _Slots(block_1=_resnet(block_in, block_in), attn_1=_attn_block(block_in),
                          block_2=_resnet(block_in, block_in))
        self.norm_out = _norm_params(block_in)
        self.z_out = 2 * z_channels if double_z else z_channels
        self.conv_out = _conv_params(block_in, self.z_out, 3)
        self._final_ch = block_in


class VectorQuantizer(nn.Module):
    """taming VectorQuantizer2 as used at sampling time (quantize.py:213-329): nearest-codebook lookup."""

    def __init__(self, n_e, e_dim, beta=0.25, remap=None, unknown_index="random", sane_index_shape=False, legacy=True):
        super().__init__()
        if remap is not None:
            raise NotImplementedError("VectorQuantizer: index remapping is not used by the shipped configs")
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.sane_index_shape = sane_index_shape
        self.embedding = _Params(weight=(n_e, e_dim))
        with torch.no_grad():
            self.embedding.weight.uniform_(-1.0 / n_e, 1.0 / n_e)

    @torch.no_grad()
    def forward(self, z, temp=None, rescale_logits=False, return_logits=False):
        """-> (z_q, loss=None, (None, None, indices)); the straight-through estimator is the identity at inference."""
        zq, idx = ops.vq_nearest(z.contiguous(), self.embedding.weight)
        idx = idx.long()
        if self.sane_index_shape:
            idx = idx.reshape(z.shape[0], z.shape[2], z.shape[3])
        return zq, None, (None, None, idx)

    @torch.no_grad()
    def get_codebook_entry(self, indices, shape):
        z_q = self.embedding.weight[indices.reshape(-1)]
        if shape is not None:
            z_q = z_q.view(shape).permute(0, 3, 1, 2).contiguous()
        return z_q


class VQModelInterface(nn.Module):
    def __init__(self, embed_dim, ddconfig=None, lossconfig=None, n_embed=None, ckpt_path=None, ignore_keys=[],
                 image_key="image", colorize_nlabels=None, monitor=None, batch_resize_range=None,
                 scheduler_config=None, lr_g_factor=1.0, remap=None, sane_index_shape=False, use_ema=False, **kw):
        super().__init__()
        ddconfig = dict(ddconfig)
        self.embed_dim, self.n_embed, self.image_key = embed_dim, n_embed, image_key
        self.ddconfig = ddconfig
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quantize = VectorQuantizer(n_embed, embed_dim, beta=0.25, remap=remap, sane_index_shape=sane_index_shape)
        self.quant_conv = _conv_params(ddconfig["z_channels"], embed_dim, 1)
        self.post_quant_conv = _conv_params(embed_dim, ddconfig["z_channels"], 1)
        if monitor is not None:
            self.monitor = monitor
        self._init_weights()
        self._packed, self._pack_sig, self._programs = None, None, {}
        self.policy_batch = None
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    @torch.no_grad()
    def _init_weights(self):
        params = dict(self.named_parameters())
        for name, p in params.items():
            if name == "quantize.embedding.weight":
                continue
            if p.dim() >= 2:
                b = 1.0 / math.sqrt(p[0].numel())
                p.uniform_(-b, b)
            elif name.endswith(".bias"):
                w = params.get(name[:-4] + "weight")
                if w is not None and w.dim() >= 2:
                    b = 1.0 / math.sqrt(w[0].numel())
                    p.uniform_(-b, b)
                else:
                    p.zero_()
            else:
                p.fill_(1.0)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")

    # ------------------------------------------------------------------------------------------
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    @torch.no_grad()
    def pack_weights(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.LdmkError("VQModelInterface: parameters must live on a GPU; there is no CPU path")
        sd = {k: v.detach().float().contiguous() for k, v in self.named_parameters()}
        P = {}
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() == 4:
                if v.shape[2] == 3:
                    narrow = v.shape[1] % 32 != 0 or v.shape[0] <= 4      # conv_in / conv_out at the latent / image boundary
                    P[k] = ops.pack_conv3x3_narrow(v) if narrow else ops.pack_conv3x3(v)
                    if not narrow and v.shape[1] >= NetBuilder.WINO_MIN_CIN and k.endswith(("conv1.weight", "conv2.weight")):
                        P[k + "#wg"] = ops.pack_winograd(v)     # ResnetBlock convs wide enough for the Winograd route
                    if not narrow and v.shape[1] >= NetBuilder.WINO_MIN_CIN and k.endswith("upsample.conv.weight"):
                        P[k + "#up"] = ops.pack_upconv(v)       # Upsample conv as four 2x2-tap phase convolutions
                elif k.startswith(("quant_conv", "post_quant_conv")):
                    P[k] = v.reshape(v.shape[0], v.shape[1]).contiguous()      # narrow NCHW 1x1: [cout][cin]
                else:
                    P[k] = ops.pack_linear(v)
        # bf16x3 images of the GEMM weights (LDMK_COMPUTE_BF16X3): used for the shapes igemm_plans_x3.json lists
        from .engine import split_enabled
        if split_enabled():
            for k in list(P):
                w = P[k]
                if k.startswith(("quant_conv", "post_quant_conv")) or not k.split("#")[0].endswith(".weight"):
                    continue
                if k.endswith(("#wg", "#up")):
                    P[k + "#s"] = ops.pack_wsplit(w, batch=w.shape[0])
                elif w.dim() == 2 and w.shape[0] % 32 == 0 and w.shape[1] % 4 == 0 and w.shape[1] > 4:
                    P[k + "#s"] = ops.pack_wsplit(w)
        self._sd, self._packed, self._pack_sig, self._programs = sd, P, self._signature(), {}

    def _ensure(self):
        if self._packed is None or self._pack_sig != self._signature():
            self.pack_weights()

    # ---- shared block emitters -----------------------------------------------------------------------
    def _resnet_block(self, nb, prefix, m, x, h, w):
        pg, P, sd, n = nb.pg, self._packed, self._sd, nb.n
        # GroupNorm+SiLU then conv: one elementwise pass + implicit GEMM, or (wide blocks, large batches) the Winograd route
        h1 = nb.gn_conv(x, None, h, w, sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], 1e-6, P[prefix + "conv1.weight"],
                        P.get(prefix + "conv1.weight#wg"), sd[prefix + "conv1.bias"], stats=True)
        g2, b2 = sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"]
        if m.cin != m.cout:
            sk = nb.lin(x.reshape(n * h * w, m.cin), P[prefix + "nin_shortcut.weight"], sd[prefix + "nin_shortcut.bias"],
                        h * w)
            out = nb.gn_conv(h1, None, h, w, g2, b2, 1e-6, P[prefix + "conv2.weight"], P.get(prefix + "conv2.weight#wg"),
                             sd[prefix + "conv2.bias"], residual=sk, out=sk.view(n, h, w, m.cout), stats=True)
        else:
            out = nb.gn_conv(h1, None, h, w, g2, b2, 1e-6, P[prefix + "conv2.weight"], P.get(prefix + "conv2.weight#wg"),
                             sd[prefix + "conv2.bias"], residual=x, stats=True)
        nb.release(h1)
        return out

    def _attn(self, nb, prefix, m, x, h, w):
        """AttnBlock, model.py:178-202: single head over h*w tokens, logits scaled by C^-0.5."""
        pg, P, sd, n = nb.pg, self._packed, self._sd, nb.n
        hw, c = h * w, m.c
        rows = n * hw
        xr = x.reshape(rows, c)
        coef = nb.gn(x, None, hw, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6)
        q = nb.lin(xr, P[prefix + "q.weight"], sd[prefix + "q.bias"], hw, tf=L.TF_AFFINE, tf_coef=coef)
        k = nb.lin(xr, P[prefix + "k.weight"], sd[prefix + "k.bias"], hw, tf=L.TF_AFFINE, tf_coef=coef)
        v = nb.lin(xr, P[prefix + "v.weight"], sd[prefix + "v.bias"], hw, tf=L.TF_AFFINE, tf_coef=coef)
        nb.release(coef)
        s = pg.alloc(n, hw, hw)
        a = ops.make_igemm_args(hw, hw, c, q, c, k, s, hw, hw, b_trans=True, batch=n, a_bstride=hw * c,
                                w_bstride=hw * c, out_bstride=hw * hw)
        pg.igemm(a, nb.pin, per_sample=True)      # one problem per sample, planned as the job's batch of them
        pg.add("ldmk_softmax_rows", s.data_ptr(), n * hw, hw, float(int(c) ** (-0.5)))
        o = pg.alloc(rows, c)
        a = ops.make_igemm_args(hw, c, hw, s, hw, v, o, c, hw, batch=n, a_bstride=hw * hw, w_bstride=hw * c,
                                out_bstride=hw * c)
        pg.igemm(a, nb.pin, per_sample=True)      # one problem per sample, planned as the job's batch of them
        nb.release(s, q, k, v)
        out = nb.lin(o, P[prefix + "proj_out.weight"], sd[prefix + "proj_out.bias"], hw, residual=xr, stats=True)
        nb.release(o)
        return out.view(n, h, w, c)

    # ---- programs -----------------------------------------------------------------------------------------
    def _build_decode(self, n, h, w, quantize):
        d = self.decoder
        P, sd = self._packed, self._sd
        dev = next(self.parameters()).device
        pg = Program(dev)
        zc = d.z_channels
        z_in = pg.alloc(n, zc, h, w)
        pg.inputs = dict(z=z_in)
        idx = pg.alloc(n * h * w, dtype=torch.int32)
        zcur = z_in
        if quantize:
            zq = pg.alloc(n, zc, h, w)
            pg.add("ldmk_vq_nearest", z_in.data_ptr(), sd["quantize.embedding.weight"].data_ptr(), zq.data_ptr(),
                   idx.data_ptr(), n, h * w, self.embed_dim, self.n_embed)
            zcur = zq
        pq = pg.alloc(n, zc, h, w)
        pg.add("ldmk_conv1x1_nchw", zcur.data_ptr(), P["post_quant_conv.weight"].data_ptr(),
               sd["post_quant_conv.bias"].data_ptr(), pq.data_ptr(), n, h * w, self.embed_dim, zc)
        top = d.ch * d.ch_mult[-1]
        hw_max = h * w * 4 ** (d.num_resolutions - 1)
        pin = None if self.policy_batch in (None, n) else (self.policy_batch, n)
        nb = NetBuilder(pg, n, pin)
        x = pg.alloc(n, h, w, top)
        pg.add("ldmk_conv3x3_in", pq.data_ptr(), zc, 0, 0, P["decoder.conv_in.weight"].data_ptr(),
               sd["decoder.conv_in.bias"].data_ptr(), x.data_ptr(), n, h, w, top)

        def step(fn, *a):
            nonlocal x
            y = fn(nb, *a[:2], x, *a[2:])
            nb.release(x)
            x = y

        step(self._resnet_block, "decoder.mid.block_1.", d.mid.block_1, h, w)
        step(self._attn, "decoder.mid.attn_1.", d.mid.attn_1, h, w)
        step(self._resnet_block, "decoder.mid.block_2.", d.mid.block_2, h, w)
        ch_, cw_ = h, w
        for lvl in reversed(range(d.num_resolutions)):
            up = d.up[lvl]
            for ib in range(d.num_res_blocks + 1):
                step(self._resnet_block, f"decoder.up.{lvl}.block.{ib}.", up.block[ib], ch_, cw_)
                if len(up.attn) > 0:
                    step(self._attn, f"decoder.up.{lvl}.attn.{ib}.", up.attn[ib], ch_, cw_)
            if lvl != 0:
                kw_ = f"decoder.up.{lvl}.upsample.conv.weight"
                y = nb.up_conv(x, ch_, cw_, P[kw_], P.get(kw_ + "#up"), sd[f"decoder.up.{lvl}.upsample.conv.bias"], stats=True)
                nb.release(x)
                x = y
                ch_, cw_ = 2 * ch_, 2 * cw_
        coef = nb.gn(x, None, ch_ * cw_, sd["decoder.norm_out.weight"], sd["decoder.norm_out.bias"], 1e-6)
        img = pg.alloc(n, d.out_ch, ch_, cw_)
        pg.add("ldmk_conv3x3_out", x.data_ptr(), coef.data_ptr(), P["decoder.conv_out.weight"].data_ptr(),
               sd["decoder.conv_out.bias"].data_ptr(), img.data_ptr(), n, ch_, cw_, d._final_ch, d.out_ch)
        pg.outputs = dict(image=img, indices=idx)
        return pg

    def _build_encode(self, n, H, W_):
        e = self.encoder
        P, sd = self._packed, self._sd
        dev = next(self.parameters()).device
        pg = Program(dev)
        x_in = pg.alloc(n, e.in_channels, H, W_)
        pg.inputs = dict(x=x_in)
        pin = None if self.policy_batch in (None, n) else (self.policy_batch, n)
        nb = NetBuilder(pg, n, pin)
        x = pg.alloc(n, H, W_, e.ch)
        pg.add("ldmk_conv3x3_in", x_in.data_ptr(), e.in_channels, 0, 0, P["encoder.conv_in.weight"].data_ptr(),
               sd["encoder.conv_in.bias"].data_ptr(), x.data_ptr(), n, H, W_, e.ch)

        def step(fn, *a):
            nonlocal x
            y = fn(nb, *a[:2], x, *a[2:])
            nb.release(x)
            x = y

        ch_, cw_ = H, W_
        for lvl in range(e.num_resolutions):
            dn = e.down[lvl]
            for ib in range(e.num_res_blocks):
                step(self._resnet_block, f"encoder.down.{lvl}.block.{ib}.", dn.block[ib], ch_, cw_)
                if len(dn.attn) > 0:
                    step(self._attn, f"encoder.down.{lvl}.attn.{ib}.", dn.attn[ib], ch_, cw_)
            if lvl != e.num_resolutions - 1:
                y = nb.conv(x, None, P[f"encoder.down.{lvl}.downsample.conv.weight"],
                            sd[f"encoder.down.{lvl}.downsample.conv.bias"], ch_, cw_, stride=2, pad_lo=0, stats=True)
                nb.release(x)
                x = y
                ch_, cw_ = (ch_ + 1 - 3) // 2 + 1, (cw_ + 1 - 3) // 2 + 1
        step(self._resnet_block, "encoder.mid.block_1.", e.mid.block_1, ch_, cw_)
        step(self._attn, "encoder.mid.attn_1.", e.mid.attn_1, ch_, cw_)
        step(self._resnet_block, "encoder.mid.block_2.", e.mid.block_2, ch_, cw_)
        coef = nb.gn(x, None, ch_ * cw_, sd["encoder.norm_out.weight"], sd["encoder.norm_out.bias"], 1e-6)
        hz = pg.alloc(n, e.z_out, ch_, cw_)
        pg.add("ldmk_conv3x3_out", x.data_ptr(), coef.data_ptr(), P["encoder.conv_out.weight"].data_ptr(),
               sd["encoder.conv_out.bias"].data_ptr(), hz.data_ptr(), n, ch_, cw_, e._final_ch, e.z_out)
        z = pg.alloc(n, self.embed_dim, ch_, cw_)
        pg.add("ldmk_conv1x1_nchw", hz.data_ptr(), P["quant_conv.weight"].data_ptr(), sd["quant_conv.bias"].data_ptr(),
               z.data_ptr(), n, ch_ * cw_, e.z_out, self.embed_dim)
        pg.outputs = dict(z=z)
        return pg

    def _program(self, kind, *key):
        self._ensure()
        k = (kind,) + key + (self.policy_batch,)
        pg = self._programs.get(k)
        if pg is None:
            pg = self._build_decode(*key) if kind == "dec" else self._build_encode(*key)
            self._programs[k] = pg
        return pg

    # ---- public surface (autoencoder.py:269-282) ------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x):
        if not x.is_cuda:
            raise L.LdmkError("VQModelInterface.encode: CUDA tensors only (no CPU fallback)")
        n, _, H, W_ = x.shape
        pg = self._program("enc", n, H, W_)
        pg.inputs["x"].copy_(x)
        pg.run()
        return pg.outputs["z"].clone()

    @torch.no_grad()
    def decode(self, h, force_not_quantize=False, return_indices=False):
        if not h.is_cuda:
            raise L.LdmkError("VQModelInterface.decode: CUDA tensors only (no CPU fallback)")
        n, _, hh, ww = h.shape
        pg = self._program("dec", n, hh, ww, not force_not_quantize)
        pg.inputs["z"].copy_(h)
        pg.run()
        img = pg.outputs["image"].clone()
        return (img, pg.outputs["indices"].clone()) if return_indices else img

    def forward(self, x):
        return self.decode(self.encode(x))
