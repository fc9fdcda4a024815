"""UNet training step on the HIP kernels (SURVEY §8f "next" row N1).

Mirrors `LatentDiffusion.p_losses` (ddpm.py:1014-1047: q_sample -> apply_model -> l2 loss .mean()), the backward
pass autograd would run through `UNetModel.forward` (openaimodel.py:710-742), `configure_optimizers` (AdamW,
ddpm.py:1363-1385) and `LitEma` (ema.py:25-44).  fp32 throughout.

Design: the forward walks the same module list as the sampling program (unet.py) but keeps what the backward needs
(normalised GEMM inputs, pre-activations, statistics) and pushes one closure per layer on a tape; the backward pops
them.  Parameters live in ONE flat fp32 buffer in the *packed* layouts the GEMMs read ([Cin/32][9][32][Cout] for 3x3
convolutions, [in][out] for Linear), gradients in a second flat buffer of the same layout -- the optimizer and the
data-parallel all-reduce see two contiguous arrays, and no repacking happens between steps.  Every launch goes to
libldmk.so on the current stream, so a whole step can be captured in a hipGraph.

Scope: cross-attention context of 1 token (both shipped configs: (B,1,512) FR, (B,1,1024) TF; the block collapses to a
per-sample vector) or up to 128 tokens (general cross-attention forward/backward kernels), dropout 0.
"""
import os

import torch
import torch.nn.functional as F

from . import lib as L
from . import switches
from . import ops
from . import train_ops as T
from .engine import tuned_plan

_SK_WS = {}
_SK_CNT = {}
RECORD = None      # tools/autotune.py --train: list collecting one (args, keepalive tensors) entry per distinct GEMM shape


def _splitk_ws(dev, elems=64 * 1024 * 1024):
    ws = _SK_WS.get(dev)
    if ws is None or ws.numel() < elems:
        ws = torch.empty(elems, device=dev, dtype=torch.float32)
        _SK_WS[dev] = ws
    return ws


# bf16 step: the forward GEMMs read their weights from bf16, transposed, K-contiguous images (ldmk_pack_wbf16t) instead of
# converting the fp32 [K][N] matrix fragment by fragment inside the kernel (eight ds_read_b32 + conversions per MFMA operand).
# A weight is registered the first time a forward GEMM meets it; repack_bf16_weights() refreshes every registered image and
# is the first thing of a forward pass (the optimiser writes the fp32 masters through raw pointers, so there is no version to
# watch: the images are simply rebuilt every step -- 0.94 GB of traffic for the 157 M parameters of this UNet).
# The registry belongs to ONE trainer (UNetTrainer._wt16, made current by its forward / backward) and only holds weights that
# live inside that trainer's flat parameter buffer: it dies with the trainer, a second trainer never repacks (or reads) the
# first one's addresses, and temporaries (mirrored-tap data-gradient weights, another model's frozen weights) never enter it.
class Bf16Images:
    def __init__(self, flat):
        self.flat = flat                 # keeps the owner's parameter buffer alive as long as its images exist
        self.lo = flat.data_ptr()
        self.hi = self.lo + flat.numel() * flat.element_size()
        self.images = {}                 # device pointer of a packed fp32 weight [K][ldb] -> (bf16 image [N][ld], ld, K, N, ldb)

    def owns(self, ptr):
        return self.lo <= ptr < self.hi


_WT16 = None        # the current trainer's Bf16Images (None: no pre-packed images, the GEMM converts in the kernel)


def activate_bf16_images(reg):
    global _WT16
    _WT16 = reg


def repack_bf16_weights():
    if T.COMPUTE != L.COMPUTE_BF16 or _WT16 is None:
        return
    st = ops.stream()
    for wptr, (img, ld, K, N, ldb) in _WT16.images.items():
        L.call("ldmk_pack_wbf16t", wptr, K, N, ldb, img.data_ptr(), ld, st)


def _bf16_image(a, dev):
    """The pre-packed bf16 image of this forward GEMM's weight, or None on first sight (it is registered for the next pass)."""
    if _WT16 is None or a.b_trans or a.batch > 1 or a.K % 8 or switches.get("LDMK_TRAIN_NO_PACKED_W") or not _WT16.owns(a.w):
        return None
    hit = _WT16.images.get(a.w)
    if hit is not None and hit[2:] == (a.K, a.N, a.ldb):
        return hit
    if torch.cuda.is_current_stream_capturing():
        return None                                     # never allocate / register inside a capture: the next eager pass will
    ld = (a.K + 7) // 8 * 8
    _WT16.images[a.w] = (torch.zeros(a.N, ld, device=dev, dtype=torch.bfloat16), ld, a.K, a.N, a.ldb)
    return None


def gemm(a, dev):
    """ldmk_igemm with the tuned (tile, split-K) plan when the shape is in the table, a shared split-K scratch."""
    ws = _splitk_ws(dev)
    if RECORD is not None:
        RECORD.append(a)
    a.compute = T.COMPUTE
    if T.COMPUTE == L.COMPUTE_BF16:
        hit = _bf16_image(a, dev)
        if hit is not None:
            a.w_split, a.w_split_ld = hit[0].data_ptr(), hit[1]
    plan = tuned_plan(a, a.M) if a.batch <= 1 else None
    if plan is not None and plan[0] > 6:
        plan = None                 # row-GEMM tiles need the fragment-order weight copy, which the trainer does not keep
    if plan is not None and max(1, a.batch) * plan[1] * a.M * a.N <= ws.numel():
        a.tile_cfg, a.splitk = plan
    a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
    if switches.get("LDMK_SPLITK_IN_LAUNCH"):       # off by default: slower than the reduce launch (engine.Program.igemm)
        cnt = _SK_CNT.get(dev)
        if cnt is None:
            cnt = _SK_CNT[dev] = torch.zeros(16384, device=dev, dtype=torch.int32)
        a.splitk_counters, a.splitk_counters_len = cnt.data_ptr(), cnt.numel()
    ops.igemm(a)


def plan_buckets(offsets, total, bucket_elems):
    """Which tail slices of the flat gradient buffer become final after which backward closure.
    offsets[i] = flat offset of the first parameter owned by the i-th closure IN EXECUTION ORDER (the tape reversed);
    layers own contiguous ranges in forward order, so after closure i everything from min(offsets[:i+1]) upwards is
    final.  Returns {closure index: (lo, hi)}: the slice to hand to the all-reduce right after that closure; slices are
    at least bucket_elems long (except the last), disjoint, and cover [0, total)."""
    out, hi, lo = {}, total, total
    for i, off in enumerate(offsets):
        last = i + 1 == len(offsets)
        lo = 0 if last else min(lo, off)
        if any(o > lo for o in offsets[i + 1:]):
            raise ValueError("backward tape is not in reverse parameter order")
        if (last or hi - lo >= bucket_elems) and hi > lo:
            out[i] = (lo, hi)
            hi = lo
    return out


class FlatParams:
    """name -> view into one flat parameter buffer (+ the matching gradient / AdamW-moment buffers)."""

    def __init__(self):
        self.specs, self.off = [], 0

    def add(self, name, tensor):
        n = tensor.numel()
        pad = (-n) % 64                      # keep every view 256-B aligned
        self.specs.append((name, tuple(tensor.shape), self.off, n, tensor))
        self.off += n + pad

    def finalize(self, dev):
        self.flat = torch.zeros(self.off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.off, device=dev, dtype=torch.float32)
        self.p, self.g = {}, {}
        for name, shape, off, n, src in self.specs:
            self.p[name] = self.flat[off:off + n].view(shape)
            self.g[name] = self.grad[off:off + n].view(shape)
            self.p[name].copy_(src)
        self.specs = [(n_, s, o, k) for n_, s, o, k, _ in self.specs]
        self.m = self.v = None
        self.step = 0


class UNetTrainer:
    def __init__(self, unet, compute="f32"):
        """compute: "f32" (the parity path: fp32 matrix cores, bit-exact fmaf-chain products) or "bf16" (BASELINE configs[4]:
        every Conv2d / Linear forward, data-gradient and weight-gradient GEMM rounds its operands to bf16 and runs on
        v_mfma_f32_32x32x16_bf16 with fp32 accumulation; master weights, gradients, AdamW, EMA, normalisations and the
        attention kernels stay fp32)."""
        self.compute = T.set_compute(compute)
        if unet._packed is None:
            unet.pack_weights()
        if unet.context_dim is None:
            raise NotImplementedError("UNetTrainer: spatial-transformer UNets with a context only")
        if not getattr(unet, "_heads32", True):
            raise NotImplementedError("UNetTrainer: attention heads of width 32 only (the flash backward kernels)")
        if getattr(unet, "use_scale_shift_norm", False) or unet.num_classes is not None:
            raise NotImplementedError("UNetTrainer: use_scale_shift_norm / num_classes UNets are built for sampling only")
        if getattr(unet, "resblock_updown", False):
            raise NotImplementedError("UNetTrainer: resblock_updown UNets are built for sampling only")
        self.unet = unet
        self.dev = next(unet.parameters()).device
        self.P = FlatParams()
        self._collect()
        self.P.finalize(self.dev)
        self._wt16 = Bf16Images(self.P.flat)      # bf16 forward-weight images: this trainer's, freed with it
        self.freqs = unet._packed["freqs"]
        self.tape = []
        self.G = {}            # activation data_ptr -> (grad tensor)
        self.ginit = set()
        self._stats = {}       # activation data_ptr -> (tensor, GroupNorm partial records emitted by its producer GEMM)
        self._first_off = {}
        self.acc_params = False   # True: backward() adds to P.grad (several passes per step: differentiable DDIM, N2)
        self.want_dx = False      # True: backward() also returns the gradient w.r.t. the network input

    # ---- parameters -----------------------------------------------------------------------------------------
    def _collect(self):
        u, P, sd, add = self.unet, self.unet._packed, self.unet._sd, self.P.add
        for k in ("te0", "te2", "emb_all", "emb_all_b"):
            add(k, P[k])
        for k in ("time_embed.0.bias", "time_embed.2.bias"):
            add(k, sd[k])
        cin = u.in_channels
        assert cin <= 32 and u.out_channels <= 32
        add("in.wpad", ops.pack_conv3x3(F.pad(sd["input_blocks.0.0.weight"], (0, 0, 0, 0, 0, 32 - cin)).contiguous()))
        add("input_blocks.0.0.bias", sd["input_blocks.0.0.bias"])
        # the flat buffer follows the forward order of the layers: the backward finishes parameter gradients from the
        # END of the buffer towards its start, so "everything past offset X is final" is a contiguous tail -- that is
        # what lets backward() hand finished buckets to the all-reduce while earlier layers are still being computed
        for prefix, m in u._walk():
            if m.kind == "res":
                for k in ("c1", "c2"):
                    add(prefix + k, P[prefix + k])
                for k in ("in_layers.0.weight", "in_layers.0.bias", "in_layers.2.bias", "out_layers.0.weight",
                          "out_layers.0.bias", "out_layers.3.bias"):
                    add(prefix + k, sd[prefix + k])
                if m.cin != m.cout:
                    add(prefix + "skip", P[prefix + "skip"])
                    add(prefix + "skip_connection.bias", sd[prefix + "skip_connection.bias"])
            elif m.kind == "st":
                for k in ("pin", "pout"):
                    add(prefix + k, P[prefix + k])
                for k in ("norm.weight", "norm.bias", "proj_in.bias", "proj_out.bias"):
                    add(prefix + k, sd[prefix + k])
                for d in range(m.depth):
                    q = f"{prefix}transformer_blocks.{d}."
                    for k in ("qkv", "o1", "q2", "k2", "v2", "o2", "ff2"):
                        add(q + k, P[q + k])
                    add(q + "ff1n", ops.pack_linear(sd[q + "ff.net.0.proj.weight"]))
                    for k in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "norm3.weight", "norm3.bias", "attn1.to_out.0.bias",
                              "attn2.to_out.0.bias", "ff.net.0.proj.bias", "ff.net.2.bias"):
                        add(q + k, sd[q + k])
            elif m.kind in ("down", "up"):
                add(prefix + "w", P[prefix + "w"])
                bk = "op.bias" if m.kind == "down" else "conv.bias"
                add(prefix + bk, sd[prefix + bk])
        for k in ("out.0.weight", "out.0.bias"):
            add(k, sd[k])
        add("out.wpad", ops.pack_conv3x3(F.pad(sd["out.2.weight"], (0, 0, 0, 0, 0, 0, 0, 32 - u.out_channels)).contiguous()))
        add("out.bpad", F.pad(sd["out.2.bias"], (0, 32 - u.out_channels)))

    def _push(self, fn, first_param_prefix):
        """Record a backward closure together with the flat offset of the first parameter it (or its layer) owns."""
        off = self._first_off.get(first_param_prefix)
        if off is None:
            off = min(o for n_, _, o, _ in self.P.specs if n_.startswith(first_param_prefix))
            self._first_off[first_param_prefix] = off
        self.tape.append((fn, off))

    # ---- gradient bookkeeping for activations -------------------------------------------------------------------
    def _grad(self, t):
        """(gradient buffer of activation t, already-written flag); marks it written."""
        k = t.data_ptr()
        g = self.G.get(k)
        if g is None:
            g = torch.empty_like(t)
            self.G[k] = g
        acc = k in self.ginit
        self.ginit.add(k)
        return g, acc

    def _alias_grad(self, t, g):
        self.G[t.data_ptr()] = g
        self.ginit.add(t.data_ptr())

    def _take(self, t):
        g = self.G.get(t.data_ptr())
        assert g is not None and t.data_ptr() in self.ginit, "gradient requested before any consumer wrote it"
        return g

    # ---- layer primitives (forward; each pushes its backward) -------------------------------------------------------
    def _lin(self, x2d, wname, bname, rows_per_sample, residual=None, batch_vec=None, x1=None, stats=False):
        p = self.P.p
        w = p[wname]
        M, N = x2d.shape[0], w.shape[1]
        out = torch.empty(M, N, device=self.dev)
        c0 = x2d.shape[1]
        c1 = 0 if x1 is None else x1.shape[1]
        a = ops.make_igemm_args(M, N, c0 + c1, x2d, c0, w, out, N, rows_per_sample, a1=x1, c1=c1,
                                bias=None if bname is None else p[bname], residual=residual,
                                batch_vec=batch_vec, batch_vec_ld=0 if batch_vec is None else batch_vec.stride(0))
        self._emit_stats(a, out, rows_per_sample, stats)
        gemm(a, self.dev)
        return out

    def _emit_stats(self, a, out, rows_per_sample, stats):
        """Let the GEMM epilogue write the GroupNorm partial records of its output (saves a statistics pass)."""
        if stats and a.M % 32 == 0 and rows_per_sample % 32 == 0:
            part = torch.empty(a.M // 32, a.N, 3, device=self.dev)
            a.stats_out = part.data_ptr()
            self._stats[out.data_ptr()] = (out, part)      # holding `out` keeps its address from being recycled

    def _partial(self, x, n, hw):
        hit = self._stats.get(x.data_ptr())
        if hit is not None:
            return hit[1]
        c = x.shape[-1]
        part = torch.empty(n * L.load().ldmk_gn_chunks(hw) * c * 3, device=self.dev)
        L.call("ldmk_gn_partial", x.data_ptr(), c, n, hw, part.data_ptr(), ops.stream())
        self._stats[x.data_ptr()] = (x, part)
        return part

    def _lin_bwd(self, dy, x2d, wname, bname, x1=None, need_dx=True):
        """Parameter gradients of out = [x2d|x1] @ W + b and (optionally) the data gradient dy @ W^T."""
        g, p = self.P.g, self.P.p
        c0 = x2d.shape[1]
        T.wgrad_linear(x2d, dy, dw=g[wname][:c0], dbias=None if bname is None else g[bname], accumulate=self.acc_params)
        if x1 is not None:
            T.wgrad_linear(x1, dy, dw=g[wname][c0:], accumulate=self.acc_params)
        if not need_dx:
            return None
        return self._lin_dx(dy, p[wname][:c0] if x1 is not None else p[wname])

    def _lin_dx(self, dy, w, out=None, residual=None):
        """dy[M][N] @ w[K][N]^T -> [M][K] (b_trans igemm on the forward weights)."""
        M, N = dy.shape
        K = w.shape[0]
        if out is None:
            out = torch.empty(M, K, device=self.dev)
        a = ops.make_igemm_args(M, K, N, dy, N, w, out, K, M, b_trans=True, ldb=w.stride(0), residual=residual)
        gemm(a, self.dev)
        return out

    def _conv(self, x4, wname, bname, stride=1, upsample=False, batch_vec=None, residual=None, stats=True):
        p = self.P.p
        n, h, w_, c = x4.shape
        wp = p[wname]
        cout = wp.shape[1]
        oh, ow = (2 * h, 2 * w_) if upsample else ((h - 1) // stride + 1, (w_ - 1) // stride + 1)
        out = torch.empty(n, oh, ow, cout, device=self.dev)
        a = ops.make_igemm_args(n * oh * ow, cout, 9 * c, x4, c, wp, out, cout, oh * ow,
                                conv=(h, w_, oh, ow, stride, 1, 1 if upsample else 0), bias=p[bname], residual=residual,
                                batch_vec=batch_vec, batch_vec_ld=0 if batch_vec is None else batch_vec.stride(0))
        self._emit_stats(a, out, oh * ow, stats)
        gemm(a, self.dev)
        return out

    def _conv_bwd(self, dy4, x4, wname, bname, stride=1, upsample=False, need_dx=True, dx_out=None, dx_acc=False):
        g, p = self.P.g, self.P.p
        n, oh, ow, cout = dy4.shape
        _, h, w_, c = x4.shape
        T.wgrad_conv3x3(x4, dy4, stride=stride, upsample=upsample, dw=g[wname], dbias=g[bname], accumulate=self.acc_params)
        if not need_dx:
            return None
        wd = T.pack_dgrad3x3(p[wname], c, cout)
        if upsample:
            du = T.conv3x3_dgrad(dy4, wd, (2 * h, 2 * w_))
            out = dx_out if dx_out is not None else torch.empty(n, h, w_, c, device=self.dev)
            return T.sumpool2(du, out=out, accumulate=dx_acc)
        if dx_out is None:
            return T.conv3x3_dgrad(dy4, wd, (h, w_), stride=stride)
        return T.conv3x3_dgrad(dy4, wd, (h, w_), stride=stride, out=dx_out, residual=dx_out if dx_acc else None)

    def _gn(self, x0, x1, hw, gname, bname, eps, silu):
        """GroupNorm(32)(+SiLU) of (the concat of) NHWC tensors, materialised; returns (y2d, saved)."""
        p = self.P.p
        n = x0.shape[0]
        c0 = x0.shape[-1]
        c1 = 0 if x1 is None else x1.shape[-1]
        pa = self._partial(x0, n, hw)
        pb = None if x1 is None else self._partial(x1, n, hw)
        coef = torch.empty(n, 2, c0 + c1, device=self.dev)
        L.call("ldmk_gn_finalize", pa.data_ptr(), c0, ops._ptr(pb), c1, n, hw, 32, eps, p[gname].data_ptr(), p[bname].data_ptr(),
               coef.data_ptr(), ops.stream())
        mr = T.gn_group_stats(pa, c0, pb, c1, n, hw, 32, eps)
        y = ops.gn_apply(x0, x1, coef, n, hw, silu=silu)
        return y, (coef, mr)

    def _gn_bwd(self, dy2d, x0, x1, hw, saved, gname, bname, silu):
        coef, mr = saved
        g, p = self.P.g, self.P.p
        n = x0.shape[0]
        dx0, a0 = self._grad(x0)
        dx1, a1 = (None, False) if x1 is None else self._grad(x1)
        T.gn_bwd(x0, x1, dy2d, coef, mr, p[gname], n, hw, silu=silu, dx0=dx0, acc0=a0, dx1=dx1, acc1=a1,
                 dgamma=g[gname], dbeta=g[bname], acc_params=self.acc_params)

    # ---- blocks -------------------------------------------------------------------------------------------
    def _res_block(self, prefix, m, x0, x1, h, w, emb_all, d_emb_all, emb_off):
        n, hw = x0.shape[0], h * w
        y1, s1 = self._gn(x0, x1, hw, prefix + "in_layers.0.weight", prefix + "in_layers.0.bias", 1e-5, True)
        y1 = y1.view(n, h, w, m.cin)
        bv = emb_all[:, emb_off:emb_off + m.cout]
        h1 = self._conv(y1, prefix + "c1", prefix + "in_layers.2.bias", batch_vec=bv)
        y2, s2 = self._gn(h1, None, hw, prefix + "out_layers.0.weight", prefix + "out_layers.0.bias", 1e-5, True)
        y2 = y2.view(n, h, w, m.cout)
        if m.cin != m.cout:
            x0r = x0.reshape(n * hw, -1)
            x1r = None if x1 is None else x1.reshape(n * hw, -1)
            skip = self._lin(x0r, prefix + "skip", prefix + "skip_connection.bias", hw, x1=x1r)
            out = self._conv(y2, prefix + "c2", prefix + "out_layers.3.bias", residual=skip)
        else:
            assert x1 is None
            out = self._conv(y2, prefix + "c2", prefix + "out_layers.3.bias", residual=x0)

        def bwd():
            dout = self._take(out)
            dy2 = self._conv_bwd(dout, y2, prefix + "c2", prefix + "out_layers.3.bias")
            dh1 = torch.empty_like(h1)
            self.G[h1.data_ptr()] = dh1
            self._gn_bwd(dy2.view(n * hw, -1), h1, None, hw, s2, prefix + "out_layers.0.weight", prefix + "out_layers.0.bias", True)
            del dy2
            T.colsum(dh1.view(n * hw, -1), rows_per_group=hw, out=d_emb_all[:, emb_off:emb_off + m.cout])
            dy1 = self._conv_bwd(dh1, y1, prefix + "c1", prefix + "in_layers.2.bias")
            self._gn_bwd(dy1.view(n * hw, -1), x0, x1, hw, s1, prefix + "in_layers.0.weight", prefix + "in_layers.0.bias", True)
            dx0 = self._take(x0)
            d2 = dout.view(n * hw, -1)
            if m.cin != m.cout:
                g, p = self.P.g, self.P.p
                g[prefix + "skip_connection.bias"].copy_(g[prefix + "out_layers.3.bias"])   # same column sums of dout
                c0 = x0.shape[-1]
                T.wgrad_linear(x0.view(n * hw, -1), d2, dw=g[prefix + "skip"][:c0], accumulate=self.acc_params)
                self._lin_dx(d2, p[prefix + "skip"][:c0], out=dx0.view(n * hw, -1), residual=dx0.view(n * hw, -1))
                if x1 is not None:
                    dx1 = self._take(x1)
                    T.wgrad_linear(x1.view(n * hw, -1), d2, dw=g[prefix + "skip"][c0:], accumulate=self.acc_params)
                    self._lin_dx(d2, p[prefix + "skip"][c0:], out=dx1.view(n * hw, -1), residual=dx1.view(n * hw, -1))
            else:
                T.axpy_(dx0, dout, 1.0)
        self._push(bwd, prefix)
        return out

    def _spatial_tf(self, prefix, m, x, h, w, ctx, dctx, L_ctx):
        n, hw = x.shape[0], h * w
        C_ = m.heads * m.d_head
        rows = n * hw
        xn, sx = self._gn(x, None, hw, prefix + "norm.weight", prefix + "norm.bias", 1e-6, False)
        hcur = self._lin(xn, prefix + "pin", prefix + "proj_in.bias", hw)
        blocks = []
        for d in range(m.depth):
            q = f"{prefix}transformer_blocks.{d}."
            p = self.P.p
            B = dict(q=q, hin=hcur)
            B["st1"] = ops.ln_stats(hcur)
            B["ln1"] = T.ln_apply(hcur, B["st1"], p[q + "norm1.weight"], p[q + "norm1.bias"])
            B["qkv"] = self._lin(B["ln1"], q + "qkv", None, hw)
            B["att"], B["lse"] = T.attn_self_lse(B["qkv"], n, hw, m.heads)
            if L_ctx == 1:
                # single context token (K11): softmax over one key is 1 -> the block adds to_out(to_v(ctx)) per sample
                B["v"] = self._lin(ctx, q + "v2", None, 1)
                cvec = self._lin(B["v"], q + "o2", q + "attn2.to_out.0.bias", 1)
                h1 = self._lin(B["att"], q + "o1", q + "attn1.to_out.0.bias", hw, residual=hcur, batch_vec=cvec)
            else:
                B["h1a"] = self._lin(B["att"], q + "o1", q + "attn1.to_out.0.bias", hw, residual=hcur)
                B["k2"] = self._lin(ctx, q + "k2", None, L_ctx)
                B["v2"] = self._lin(ctx, q + "v2", None, L_ctx)
                B["st2"] = ops.ln_stats(B["h1a"])
                B["ln2"] = T.ln_apply(B["h1a"], B["st2"], p[q + "norm2.weight"], p[q + "norm2.bias"])
                B["q2"] = self._lin(B["ln2"], q + "q2", None, hw)
                B["a2"] = ops.attn_cross(B["q2"], B["k2"], B["v2"], n, hw, L_ctx, m.heads)
                h1 = self._lin(B["a2"], q + "o2", q + "attn2.to_out.0.bias", hw, residual=B["h1a"])
            B["h1"] = h1
            B["st3"] = ops.ln_stats(h1)
            B["ln3"] = T.ln_apply(h1, B["st3"], p[q + "norm3.weight"], p[q + "norm3.bias"])
            B["pre"] = self._lin(B["ln3"], q + "ff1n", q + "ff.net.0.proj.bias", hw)
            B["f"] = T.geglu_fwd(B["pre"])
            hcur = self._lin(B["f"], q + "ff2", q + "ff.net.2.bias", hw, residual=h1)
            blocks.append(B)
        h_last = hcur
        out = self._lin(h_last, prefix + "pout", prefix + "proj_out.bias", hw, residual=x.view(rows, m.ch),
                        stats=True).view(n, h, w, m.ch)

        def add_dctx(dy, wname):
            first = not self._dctx_init
            self._lin_dx(dy, self.P.p[wname], out=dctx, residual=None if first else dctx)
            self._dctx_init = True

        def bwd():
            g, p = self.P.g, self.P.p
            acc = self.acc_params
            dout = self._take(out).view(rows, m.ch)
            dh = self._lin_bwd(dout, h_last, prefix + "pout", prefix + "proj_out.bias")
            for B in reversed(blocks):
                q = B["q"]
                # ---- feed-forward: h2 = h1 + ff2(geglu(ff1(LN3(h1))))
                df = self._lin_bwd(dh, B["f"], q + "ff2", q + "ff.net.2.bias")
                dpre = T.geglu_bwd(B["pre"], df)
                del df
                dln3 = self._lin_bwd(dpre, B["ln3"], q + "ff1n", q + "ff.net.0.proj.bias")
                del dpre
                T.ln_bwd(dln3, B["h1"], B["st3"], p[q + "norm3.weight"], dx=dh, acc_dx=True, dgamma=g[q + "norm3.weight"],
                         dbeta=g[q + "norm3.bias"], acc_params=acc)                 # dh is now d(h1)
                del dln3
                if L_ctx == 1:
                    # ---- h1 = hin + to_out(attn(LN1(hin))) + cvec[sample]
                    dcvec = T.colsum(dh, rows_per_group=hw)
                    datt = self._lin_bwd(dh, B["att"], q + "o1", q + "attn1.to_out.0.bias")
                    dv = self._lin_bwd(dcvec, B["v"], q + "o2", q + "attn2.to_out.0.bias")
                    T.wgrad_linear(ctx, dv, dw=g[q + "v2"], accumulate=acc)
                    add_dctx(dv, q + "v2")
                else:
                    # ---- h1 = h1a + to_out2(cross_attn(LN2(h1a), ctx));  h1a = hin + to_out(attn(LN1(hin)))
                    da2 = self._lin_bwd(dh, B["a2"], q + "o2", q + "attn2.to_out.0.bias")
                    dq2, dk2, dv2 = T.attn_cross_bwd(B["q2"], B["k2"], B["v2"], da2, n, hw, L_ctx, m.heads)
                    del da2
                    dln2 = self._lin_bwd(dq2, B["ln2"], q + "q2", None)
                    T.ln_bwd(dln2, B["h1a"], B["st2"], p[q + "norm2.weight"], dx=dh, acc_dx=True, dgamma=g[q + "norm2.weight"],
                             dbeta=g[q + "norm2.bias"], acc_params=acc)             # dh is now d(h1a)
                    del dln2, dq2
                    T.wgrad_linear(ctx, dk2, dw=g[q + "k2"], accumulate=acc)
                    T.wgrad_linear(ctx, dv2, dw=g[q + "v2"], accumulate=acc)
                    add_dctx(dk2, q + "k2")
                    add_dctx(dv2, q + "v2")
                    datt = self._lin_bwd(dh, B["att"], q + "o1", q + "attn1.to_out.0.bias")
                dqkv = T.attn_self_bwd(B["qkv"], B["att"], datt, B["lse"], n, hw, m.heads)   # flash style: no [T][T] matrix
                del datt
                dln1 = self._lin_bwd(dqkv, B["ln1"], q + "qkv", None)
                del dqkv
                T.ln_bwd(dln1, B["hin"], B["st1"], p[q + "norm1.weight"], dx=dh, acc_dx=True, dgamma=g[q + "norm1.weight"],
                         dbeta=g[q + "norm1.bias"], acc_params=acc)                 # dh is now d(hin)
                del dln1
            dxn = self._lin_bwd(dh, xn, prefix + "pin", prefix + "proj_in.bias")
            self._gn_bwd(dxn, x, None, hw, sx, prefix + "norm.weight", prefix + "norm.bias", False)
            T.axpy_(self._take(x), dout.view_as(x), 1.0)
        self._push(bwd, prefix)
        return out

    def _down(self, prefix, x, h, w):
        out = self._conv(x, prefix + "w", prefix + "op.bias", stride=2)

        def bwd():
            dx, acc = self._grad(x)
            self._conv_bwd(self._take(out), x, prefix + "w", prefix + "op.bias", stride=2, dx_out=dx, dx_acc=acc)
        self._push(bwd, prefix)
        return out

    def _up(self, prefix, x, h, w):
        out = self._conv(x, prefix + "w", prefix + "conv.bias", upsample=True)

        def bwd():
            dx, acc = self._grad(x)
            self._conv_bwd(self._take(out), x, prefix + "w", prefix + "conv.bias", upsample=True, dx_out=dx, dx_acc=acc)
        self._push(bwd, prefix)
        return out

    # ---- whole network -------------------------------------------------------------------------------------
    def forward(self, x, timesteps, context):
        """x (n,C_in,H,W) fp32 NCHW (already concatenated with any c_concat), timesteps (n,) int64, context (n,L,ctx_dim)
        (L = 1: the shipped configs' fast path; 1 < L <= 128: general cross-attention).
        Returns eps (n,C_out,H,W); records the tape for backward()."""
        u, p, dev = self.unet, self.P.p, self.dev
        T.set_compute(self.compute)
        if not x.is_cuda:
            raise L.LdmkError("UNetTrainer.forward: CUDA tensors only (no CPU fallback)")
        activate_bf16_images(self._wt16)
        repack_bf16_weights()            # (bf16 step only) the forward weights' bf16 images follow the optimiser's last update
        if context is None:
            raise L.LdmkError("UNetTrainer.forward: context is required")
        L_ctx = context.shape[1]
        if L_ctx > 128:
            raise NotImplementedError("UNetTrainer: context longer than 128 tokens")
        n, cin, H, W_ = x.shape
        self.tape, self.G, self.ginit, self._stats = [], {}, set(), {}
        self._dctx_init = False
        mc = u.model_channels
        ctx = context.reshape(n * L_ctx, u.context_dim).contiguous().float()
        self.dctx = torch.zeros_like(ctx)
        xp = torch.zeros(n, H, W_, 32, device=dev)
        xp[..., :cin] = x.permute(0, 2, 3, 1)
        # ---- timestep embedding MLP and every ResBlock's emb_layers in one GEMM (K1)
        temb = ops.timestep_embedding(timesteps.to(torch.int64), self.freqs, mc)
        e1 = self._lin(temb, "te0", "time_embed.0.bias", 1)
        s1 = T.silu(e1)
        emb = self._lin(s1, "te2", "time_embed.2.bias", 1)
        s2 = T.silu(emb)
        emb_all = self._lin(s2, "emb_all", "emb_all_b", 1)
        d_emb_all = torch.zeros_like(emb_all)

        def bwd_emb():
            ds2 = self._lin_bwd(d_emb_all, s2, "emb_all", "emb_all_b")
            demb = T.silu_bwd(emb, ds2)
            ds1 = self._lin_bwd(demb, s1, "te2", "time_embed.2.bias")
            de1 = T.silu_bwd(e1, ds1)
            self._lin_bwd(de1, temb, "te0", "time_embed.0.bias", need_dx=False)
        self._push(bwd_emb, "te0")

        h0 = self._conv(xp, "in.wpad", "input_blocks.0.0.bias")

        def bwd_in():
            dxp = self._conv_bwd(self._take(h0), xp, "in.wpad", "input_blocks.0.0.bias", need_dx=self.want_dx)
            if dxp is not None:
                self._dx = dxp[..., :cin].permute(0, 3, 1, 2).contiguous()
        self._push(bwd_in, "in.wpad")

        def run_layers(prefix, layers, x0, x1, h, w):
            cur0, cur1 = x0, x1
            for j, m in enumerate(layers):
                pf = f"{prefix}{j}."
                if m.kind == "res":
                    cur0 = self._res_block(pf, m, cur0, cur1, h, w, emb_all, d_emb_all, u._emb_off[pf])
                elif m.kind == "st":
                    cur0 = self._spatial_tf(pf, m, cur0, h, w, ctx, self.dctx, L_ctx)
                elif m.kind == "down":
                    cur0 = self._down(pf, cur0, h, w)
                    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                elif m.kind == "up":
                    cur0 = self._up(pf, cur0, h, w)
                    h, w = 2 * h, 2 * w
                else:
                    raise AssertionError(m.kind)
                cur1 = None
            return cur0, h, w

        hs = [(h0, H, W_)]
        hcur, ch_, cw_ = h0, H, W_
        for i in range(1, len(u.input_blocks)):
            hcur, ch_, cw_ = run_layers(f"input_blocks.{i}.", u.input_blocks[i].layers, hcur, None, ch_, cw_)
            hs.append((hcur, ch_, cw_))
        hcur, ch_, cw_ = run_layers("middle_block.", u.middle_block.layers, hcur, None, ch_, cw_)
        for i, blk in enumerate(u.output_blocks):
            skip, sh, sw = hs.pop()
            assert (sh, sw) == (ch_, cw_)
            hcur, ch_, cw_ = run_layers(f"output_blocks.{i}.", blk.layers, hcur, skip, ch_, cw_)
        hw = ch_ * cw_
        yo, so = self._gn(hcur, None, hw, "out.0.weight", "out.0.bias", 1e-5, True)
        yo = yo.view(n, ch_, cw_, -1)
        eps_pad = self._conv(yo, "out.wpad", "out.bpad", stats=False)
        h_final = hcur

        def bwd_out():
            dyo = self._conv_bwd(self._take(eps_pad), yo, "out.wpad", "out.bpad")
            self._gn_bwd(dyo.view(n * hw, -1), h_final, None, hw, so, "out.0.weight", "out.0.bias", True)
        self._push(bwd_out, "out.0.weight")
        self.eps_pad = eps_pad
        self.last_pass = dict(tape=self.tape, G=self.G, ginit=self.ginit, eps_pad=eps_pad, dctx=self.dctx)
        return eps_pad[..., :u.out_channels].permute(0, 3, 1, 2).contiguous()

    def backward(self, deps_pad, pass_=None, reduce_world=1, bucket_elems=32 * 1024 * 1024):
        """deps_pad: gradient w.r.t. the channel-padded NHWC output (n,H,W,32) of the pass `pass_` (default: the last
        forward).  Fills (or, with acc_params, adds to) P.grad, sets self.dctx; returns d(loss)/d(input) when want_dx.
        reduce_world > 1: data-parallel averaging overlapped with the backward -- as soon as the tail of the flat
        gradient buffer past some offset is final, that bucket (>= bucket_elems floats, 128 MB by default: few, large
        collectives suit the xGMI rings) is handed to an asynchronous all-reduce while earlier layers still compute."""
        ps = self.last_pass if pass_ is None else pass_
        T.set_compute(self.compute)
        activate_bf16_images(self._wt16)
        self.tape, self.G, self.ginit, self.eps_pad, self.dctx = ps["tape"], ps["G"], ps["ginit"], ps["eps_pad"], ps["dctx"]
        self._dctx_init = False
        self._alias_grad(self.eps_pad, deps_pad)
        self._stats = {}
        self._dx = None
        works = []
        order = list(reversed(self.tape))
        buckets = plan_buckets([off for _, off in order], self.P.grad.numel(), bucket_elems) if reduce_world > 1 else {}
        for i, (fn, _) in enumerate(order):
            fn()
            if i in buckets:
                import torch.distributed as dist
                lo, hi = buckets[i]
                works.append(dist.all_reduce(self.P.grad[lo:hi], async_op=True))
        for w in works:
            w.wait()
        if reduce_world > 1:
            self.P.grad.mul_(1.0 / reduce_world)
        self.tape, self.G, self.ginit = [], {}, set()
        ps["tape"] = ps["G"] = ps["ginit"] = None             # the saved activations die with the closures
        return self._dx

    @staticmethod
    def pad_output_grad(deps_nchw):
        """(n,C_out,H,W) gradient -> the channel-padded NHWC layout backward() takes."""
        n, co, H, W_ = deps_nchw.shape
        out = torch.zeros(n, H, W_, 32, device=deps_nchw.device)
        out[..., :co] = deps_nchw.permute(0, 2, 3, 1)
        return out

    # ---- p_losses / optimizer -------------------------------------------------------------------------------
    def p_losses(self, x_start, context, t, noise, sqrt_ac, sqrt_1mac, c_concat=None, reduce_world=1):
        """ddpm.py:1014-1047 with parameterization 'eps', loss_type 'l2', l_simple_weight 1, no learned logvar,
        original_elbo_weight 0: loss = mean((eps_theta(q_sample(x0,t,noise), t, c) - noise)^2); `c_concat` (masked-frame +
        identity latents of the talking-face model) is concatenated to the noisy latent on the channel axis.  Returns the loss
        (device scalar) after running forward + backward; gradients are in self.P.grad."""
        x_noisy = T.q_sample(x_start.contiguous(), noise.contiguous(), t, sqrt_ac, sqrt_1mac)
        if c_concat is not None:         # TF DiffusionWrapper: torch.cat([x] + c_concat, dim=1), ddpm2cond.py:1309
            x_noisy = torch.cat([x_noisy, c_concat.float()], 1)
        self.forward(x_noisy, t, context)
        n, co, H, W_ = noise.shape
        tgt = torch.zeros_like(self.eps_pad)
        tgt[..., :co] = noise.permute(0, 2, 3, 1)
        loss, deps = T.mse_grad(self.eps_pad, tgt, denom=noise.numel())
        self.backward(deps, reduce_world=reduce_world)
        return loss

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        P = self.P
        if P.m is None:
            P.m, P.v = torch.zeros_like(P.flat), torch.zeros_like(P.flat)
        P.step += 1
        T.adamw_(P.flat, P.grad, P.m, P.v, lr, betas, eps, weight_decay, P.step)

    def optimizer_state(self):
        """Resume point of the optimizer (SURVEY §5 checkpoint/resume): packed weights, AdamW moments, step count."""
        P = self.P
        return dict(flat=P.flat.clone(), m=None if P.m is None else P.m.clone(), v=None if P.v is None else P.v.clone(),
                    step=P.step, layout=[(n_, tuple(s_), o, k) for n_, s_, o, k in P.specs])

    def load_optimizer_state(self, st):
        P = self.P
        if [(n_, tuple(s_), o, k) for n_, s_, o, k in P.specs] != list(st["layout"]):
            raise ValueError("optimizer state was saved for a different UNet configuration (flat layout mismatch)")
        P.flat.copy_(st["flat"])
        P.m = None if st["m"] is None else st["m"].to(P.flat.device).clone()
        P.v = None if st["v"] is None else st["v"].to(P.flat.device).clone()
        P.step = int(st["step"])

    def ema_update(self, shadow_flat, decay):
        T.ema_(shadow_flat, self.P.flat, 1.0 - decay)

    def pack_reference_state(self, sd):
        """Reference-layout tensors (state-dict key -> tensor, e.g. the `model_ema` shadow of a checkpoint) -> one flat
        buffer in this trainer's packed layout: the inverse of state_dict_reference()."""
        flat = torch.zeros_like(self.P.flat)
        sd = {k: v.to(self.dev, torch.float32) for k, v in sd.items()}
        for name, shape, off, n in self.P.specs:
            flat[off:off + n].view(shape).copy_(reference_grad_layout(self.unet, name, sd).reshape(shape))
        return flat

    # ---- back to the reference's state-dict layout -------------------------------------------------------------
    def state_dict_reference(self, flat=None):
        """Unpack the flat (packed-layout) parameters into the reference's state-dict keys and shapes
        (openaimodel.py module tree); `flat` defaults to the live weights, pass an EMA shadow buffer to export that.
        Every parameter of the UNet lives in the flat buffer (attn2.to_q / to_k / norm2 receive exactly zero gradient
        with a single-token context, as in the reference)."""
        u = self.unet
        src = self.P.flat if flat is None else flat
        view = {name: src[off:off + n].view(shape) for name, shape, off, n in self.P.specs}
        out = {k: v.detach().clone() for k, v in u._sd.items()}

        def unconv(wp, cin, cout):                 # [cin/32][9][32][cout] -> [cout][cin][3][3]
            return wp.view(cin // 32, 9, 32, cout).permute(3, 0, 2, 1).reshape(cout, cin, 3, 3).contiguous()

        def unlin(wp, like):                       # [in][out] -> the reference weight's own shape ([out][in] or [out][in][1][1])
            return wp.t().contiguous().view(like.shape)

        lin = {"skip": "skip_connection.weight", "pin": "proj_in.weight", "pout": "proj_out.weight",
               "o1": "attn1.to_out.0.weight", "q2": "attn2.to_q.weight", "k2": "attn2.to_k.weight",
               "v2": "attn2.to_v.weight", "o2": "attn2.to_out.0.weight", "ff2": "ff.net.2.weight",
               "ff1n": "ff.net.0.proj.weight"}
        for name, wp in view.items():
            if name in ("te0", "te2"):
                out[f"time_embed.{name[2]}.weight"] = wp.t().contiguous()
            elif name in ("emb_all", "emb_all_b"):
                for pf, m in u._walk():
                    if m.kind == "res":
                        o = u._emb_off[pf]
                        if name == "emb_all":
                            out[pf + "emb_layers.1.weight"] = wp[:, o:o + m.cout].t().contiguous()
                        else:
                            out[pf + "emb_layers.1.bias"] = wp[o:o + m.cout].clone()
            elif name == "in.wpad":
                out["input_blocks.0.0.weight"] = unconv(wp, 32, wp.shape[1])[:, :u.in_channels].contiguous()
            elif name == "out.wpad":
                out["out.2.weight"] = unconv(wp, wp.shape[0] // 9, 32)[:u.out_channels].contiguous()
            elif name == "out.bpad":
                out["out.2.bias"] = wp[:u.out_channels].clone()
            elif name.endswith(".c1"):
                out[name[:-2] + "in_layers.2.weight"] = unconv(wp, wp.shape[0] // 9, wp.shape[1])
            elif name.endswith(".c2"):
                out[name[:-2] + "out_layers.3.weight"] = unconv(wp, wp.shape[0] // 9, wp.shape[1])
            elif name.endswith(".w"):
                base = name[:-1]
                key = base + ("op.weight" if base + "op.weight" in out else "conv.weight")
                out[key] = unconv(wp, wp.shape[0] // 9, wp.shape[1])
            elif name.endswith(".qkv"):
                b, c_ = name[:-3], wp.shape[0]
                for i, ch in enumerate("qkv"):
                    out[b + f"attn1.to_{ch}.weight"] = wp[:, i * c_:(i + 1) * c_].t().contiguous()
            else:
                for suf, key in lin.items():
                    if name.endswith("." + suf):
                        k = name[:-len(suf)] + key
                        out[k] = unlin(wp, out[k])
                        break
                else:
                    out[name] = wp.clone()
        return out

    def sync_to_module(self, flat=None):
        """Write the trained weights back into the wrapped UNetModel's nn.Parameters and re-pack its sampling program."""
        sd = self.state_dict_reference(flat)
        with torch.no_grad():
            for k, prm in self.unet.named_parameters():
                prm.copy_(sd[k])
        self.unet.pack_weights()

    def all_reduce_grads(self, world_size):
        """Data-parallel gradient averaging: one RCCL all-reduce over the flat gradient buffer (main.py:532 DDP)."""
        import torch.distributed as dist
        dist.all_reduce(self.P.grad)
        self.P.grad.mul_(1.0 / world_size)


def reference_grad_layout(unet, name, grads):
    """The gradient `name` of the flat packed layout, computed from reference-layout gradients `grads`
    (state-dict key -> tensor, e.g. from autograd on the oracle): what tests compare the HIP gradients against."""
    dev = next(iter(grads.values())).device
    cat = torch.cat
    if name in ("te0", "te2"):
        return grads[f"time_embed.{name[2]}.weight"].t()
    if name == "emb_all":
        return cat([grads[p + "emb_layers.1.weight"] for p, m in unet._walk() if m.kind == "res"], 0).t()
    if name == "emb_all_b":
        return cat([grads[p + "emb_layers.1.bias"] for p, m in unet._walk() if m.kind == "res"], 0)
    if name == "in.wpad":
        return ops.pack_conv3x3(F.pad(grads["input_blocks.0.0.weight"], (0, 0, 0, 0, 0, 32 - unet.in_channels)).contiguous())
    if name == "out.wpad":
        return ops.pack_conv3x3(F.pad(grads["out.2.weight"], (0, 0, 0, 0, 0, 0, 0, 32 - unet.out_channels)).contiguous())
    if name == "out.bpad":
        return F.pad(grads["out.2.bias"], (0, 32 - unet.out_channels))
    for suf, key in (("c1", "in_layers.2.weight"), ("c2", "out_layers.3.weight")):
        if name.endswith("." + suf):
            return ops.pack_conv3x3(grads[name[:-len(suf)] + key].contiguous())
    if name.endswith(".w"):
        base = name[:-1]
        key = base + ("op.weight" if base + "op.weight" in grads else "conv.weight")
        return ops.pack_conv3x3(grads[key].contiguous())
    lin = {"skip": "skip_connection.weight", "pin": "proj_in.weight", "pout": "proj_out.weight", "o1": "attn1.to_out.0.weight",
           "q2": "attn2.to_q.weight", "k2": "attn2.to_k.weight", "v2": "attn2.to_v.weight", "o2": "attn2.to_out.0.weight",
           "ff2": "ff.net.2.weight", "ff1n": "ff.net.0.proj.weight"}
    for suf, key in lin.items():
        if name.endswith("." + suf):
            wgt = grads[name[:-len(suf)] + key]
            return wgt.reshape(wgt.shape[0], -1).t()
    if name.endswith(".qkv"):
        b = name[:-3]
        return cat([grads[b + f"attn1.to_{c}.weight"] for c in "qkv"], 0).t()
    return grads[name]
