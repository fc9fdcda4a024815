"""Small-batch (latency-bound) launch program of the UNet: the same forward as `UNetModel._build`, re-cut for batch 1-2.

Batch 1 is the mode the reference actually ships for talking faces (`progressive_sampling`,
talking_face/progressive_sampling_difftalk.py:282-317, batch size 1 at :350): 25 600 UNet evaluations per 128-frame clip,
each a chain of dependent launches in which every separate reduce / statistics / apply pass costs a launch ramp plus a
memory round trip (tools/marginal_cost.py: 3.4 us per GroupNorm finalize, 2.7 per apply, 2.8 per LayerNorm statistics, 17.8
per GEMM with its reduce).  The batched program (`unet.py`) is cut for throughput: statistics records from GEMM epilogues,
finalize + apply passes, LayerNorm folded through the next product.  Here the cut is for FEWER DEPENDENT LAUNCHES:

  * every GEMM leaves raw split-K slabs (`raw_slabs`, slab GEMM `csrc/sgemm.hip`) and the ONE launch that follows it
    (`ldmk_post`, `csrc/post.hip`) sums them, applies bias / timestep vector / residual, stores the residual stream and
    produces the next GEMM's normalised input (GroupNorm+SiLU over the skip concat, or LayerNorm) -- no reduce, finalize,
    apply or statistics launches;
  * attention reads the QKV slabs directly (`ldmk_attn_self_small`: keys split over the waves of a workgroup);
  * `ff.net.2` and `proj_out` are one GEMM: both are linear and nothing sits between them but a residual add
    (attention.py:211-215,259-261), so  x + b_o + (h + b_2 + f W_2) W_o  =  x + (b_o + b_2 W_o) + [f | h] [W_2 W_o ; W_o]
    with the product matrix formed once in float64 at pack time;
  * the skip 1x1 convolution of a ResBlock writes its slabs next to the second 3x3 convolution's: one post sums both.

ResBlock: post, conv, post, conv (+ skip GEMM)                                = 4-5 launches (8-10 in the batched program)
SpatialTransformer: post, proj_in, post, qkv, attention, to_out, post, GEGLU proj, [post], ff2*proj_out  = 9-10 (19)
Same arithmetic up to summation order; parity tests: tests/test_unet_gpu.py (reference fixtures), tests/test_small_batch_gpu.py.
"""
import os

import torch

from . import lib as L
from . import switches
from . import ops
from .engine import Program

SMALL_ROWS = int(switches.get("LDMK_SMALL_ROWS", "4096"))


def wants_small_route(policy_n, H, W_, L_ctx):
    """Decided on the JOB's batch (policy_n), like the tile plans: a sample's result does not depend on how a batch is
    sharded.  One-token contexts only (all shipped configs); longer contexts keep the batched program."""
    return (L_ctx == 1 and policy_n * H * W_ <= SMALL_ROWS and H % 8 == 0 and W_ % 8 == 0
            and not switches.get("LDMK_NO_SMALL_ROUTE"))


@torch.no_grad()
def pack_small(unet):
    """Extra weight forms of the small-batch route (packed once per weight version, on first use): fragment-order copies
    of the UNFOLDED qkv / GEGLU projections (the post launch materialises LayerNorm(x), so the GEMM reads plain weights),
    the ff.net.2 * proj_out product matrices, the summed biases of conv2 + skip."""
    P, sd = unet._packed, unet._sd
    if P.get("#small"):
        return
    for prefix, m in unet._walk():
        if m.kind == "st":
            last = f"{prefix}transformer_blocks.{m.depth - 1}."
            for d in range(m.depth):
                q = f"{prefix}transformer_blocks.{d}."
                for k in ("qkv", "ff1"):
                    P[q + k + "#uf"] = ops.pack_wfrag(P[q + k])
            w2, wo = P[last + "ff2"].double(), P[prefix + "pout"].double()           # [4C][C], [C][ch]
            wm = torch.cat([w2 @ wo, wo], 0).float().contiguous()                     # [5C][ch]
            P[last + "ff2o"] = wm
            P[last + "ff2o#f"] = ops.pack_wfrag(wm)
            P[last + "ff2o#b"] = (sd[prefix + "proj_out.bias"].double() + sd[last + "ff.net.2.bias"].double() @ wo).float().contiguous()
        elif m.kind == "res" and m.cin != m.cout:
            P[prefix + "c2skip#b"] = (sd[prefix + "out_layers.3.bias"] + sd[prefix + "skip_connection.bias"]).contiguous()
            # the 1x1 skip connection as extra K rows of the second 3x3 convolution: [9 cout + cin][cout]
            P[prefix + "c2s"] = torch.cat([P[prefix + "c2"], P[prefix + "skip"]], 0).contiguous()
            P[prefix + "c2s#f"] = ops.pack_wfrag(P[prefix + "c2s"])
        elif m.kind in ("down", "up"):
            pass
    P["#small"] = True


class Lazy:
    """A tensor of the walk: `raw` [M][N] is where its values are (done) or will be stored by the post launch that
    consumes the pending slabs [nslab][M][N] + bias + per-sample vector + residual."""

    def __init__(self, raw, M, N, hw, slabs=None, nslab=0, bias=None, bvec=0, bvec_ld=0, residual=None):
        self.raw, self.M, self.N, self.hw = raw, M, N, hw
        self.slabs, self.nslab, self.bias, self.bvec, self.bvec_ld, self.residual = slabs, nslab, bias, bvec, bvec_ld, residual
        self.done = slabs is None


class SmallBuilder:
    def __init__(self, pg, n, pin):
        self.pg, self.n, self.pin = pg, n, pin

    # ---- GEMMs that leave raw slabs --------------------------------------------------------------------------------
    def raw(self, gemms, M, N, hw, bias=None, bvec=0, bvec_ld=0, residual=None):
        """Plan and record GEMMs whose raw products add up to one [M][N] tensor; returns its Lazy."""
        pg = self.pg
        sks = [a.splitk if getattr(a, "_planned", False) else pg.plan(a, self.pin)[1] for a in gemms]
        slabs = pg.alloc(sum(sks), M, N)
        off = 0
        for a, sk in zip(gemms, sks):
            pg.igemm_raw(a, slabs[off:off + sk])
            off += sk
        return Lazy(pg.alloc(M, N), M, N, hw, slabs=slabs, nslab=off, bias=bias, bvec=bvec, bvec_ld=bvec_ld, residual=residual)

    def _src(self, x):
        """post arguments describing where x's values come from; (kwargs, tensor to release afterwards)"""
        if x.done:
            return dict(src=x.raw, nslab=1), None
        kw = dict(src=x.slabs, nslab=x.nslab, slab_stride=x.M * x.N, bias=x.bias, batch_vec=x.bvec, batch_vec_ld=x.bvec_ld,
                  residual=x.residual, raw_out=x.raw)
        return kw, x.slabs

    def _finish(self, x, slabs, a, keep):
        self.pg.post(a, keep)
        if slabs is not None:
            self.pg.release(slabs)
            x.slabs, x.done = None, True

    def post_gn(self, x, x1, gamma, beta, eps, silu):
        """GroupNorm(32)[+SiLU] of (the channel concat of) x | x1 -> [M][N + c1]; x's raw values are stored on the way."""
        kw, slabs = self._src(x)
        c1 = 0 if x1 is None else x1.N
        assert x1 is None or x1.done
        out = self.pg.alloc(x.M, x.N + c1)
        a = ops.make_post_args(M=x.M, N=x.N, rows_per_sample=x.hw, norm=L.POST_GROUPNORM, x1=None if x1 is None else x1.raw,
                               c1=c1, gamma=gamma, beta=beta, eps=eps, silu=silu, norm_out=out, **kw)
        self._finish(x, slabs, a, (x, x1, gamma, beta, out))
        return out

    def post_ln(self, x, gamma, beta):
        kw, slabs = self._src(x)
        out = self.pg.alloc(x.M, x.N)
        a = ops.make_post_args(M=x.M, N=x.N, rows_per_sample=x.hw, norm=L.POST_LAYERNORM, gamma=gamma, beta=beta, eps=1e-5,
                               norm_out=out, **kw)
        self._finish(x, slabs, a, (x, gamma, beta, out))
        return out

    def plain(self, x):
        """Make x's raw values exist (a plain reduce + epilogue launch) when no norm launch has done it yet."""
        if not x.done:
            kw, slabs = self._src(x)
            a = ops.make_post_args(M=x.M, N=x.N, rows_per_sample=x.hw, **kw)
            self._finish(x, slabs, a, (x,))
        return x.raw


def build_small(unet, n, H, W_, L_ctx, c_concat, policy_n):
    assert L_ctx == 1
    pack_small(unet)
    # ldmk_dense_small's 16-byte-load form (<= 4 batch rows per launch): requested from the JOB's batch, so that every shard of
    # a job makes the same choice (the two forms sum K in different orders)
    ds4 = 2 if max(policy_n, n) <= 4 else 0
    P, sd = unet._packed, unet._sd
    dev = next(unet.parameters()).device
    pg = Program(dev)
    mc = unet.model_channels
    emb_ch = 4 * mc
    cx = unet.in_channels - c_concat
    x_in = pg.alloc(n, cx, H, W_)
    cc_in = pg.alloc(n, c_concat, H, W_) if c_concat else None
    t_in = pg.alloc(n, dtype=torch.int64)
    ctx_in = pg.alloc(n * L_ctx, unet.context_dim)
    pg.inputs = dict(x=x_in, c_concat=cc_in, t=t_in, context=ctx_in)
    ctx_pg = Program(dev)
    ctx_pg._all = pg._all
    p_ = lambda t: 0 if t is None else (t if isinstance(t, int) else t.data_ptr())
    pin = (policy_n, n)
    sb = SmallBuilder(pg, n, pin)

    # -- time embedding MLP + all emb_layers (one launch each), as in the batched program.  (They only read `t`; running
    # them as a parallel branch of the captured step -- a forked stream, engine.Program.side_calls -- was measured: the
    # fork / join inside the hipGraph costs more than the 45 us it hides, 2265 -> 2423 us per step.  Kept in line.)
    temb = pg.alloc(n, mc)
    pg.add("ldmk_timestep_embedding", p_(t_in), p_(P["freqs"]), p_(temb), n, mc)
    e1 = pg.alloc(n, emb_ch)
    pg.add("ldmk_dense_small", p_(temb), mc, p_(P["te0"]), p_(sd["time_embed.0.bias"]), p_(e1), emb_ch, n, mc, emb_ch, ds4)
    emb = pg.alloc(n, emb_ch)
    pg.add("ldmk_dense_small", p_(e1), emb_ch, p_(P["te2"]), p_(sd["time_embed.2.bias"]), p_(emb), emb_ch, n, emb_ch, emb_ch, 1 | ds4)
    emb_all = pg.alloc(n, unet._emb_total)
    pg.add("ldmk_dense_small", p_(emb), emb_ch, p_(P["emb_all"]), p_(P["emb_all_b"]), p_(emb_all), unet._emb_total, n,
           emb_ch, unet._emb_total, 1 | ds4)

    def conv_args(a_in, cin, wp, wf, cout, h, w, stride=1, upsample=False):
        oh, ow = (2 * h, 2 * w) if upsample else ((h - 1) // stride + 1, (w - 1) // stride + 1)
        return ops.make_igemm_args(n * oh * ow, cout, 9 * cin, a_in, cin, wp, None, cout, oh * ow,
                                   conv=(h, w, oh, ow, stride, 1, 1 if upsample else 0), w_frag=wf), oh, ow

    def res_block(prefix, m, x0, x1, h, w):
        hw, rows = h * w, n * h * w
        a1 = sb.post_gn(x0, x1, sd[prefix + "in_layers.0.weight"], sd[prefix + "in_layers.0.bias"], 1e-5, True)
        g1, _, _ = conv_args(a1, m.cin, P[prefix + "c1"], P.get(prefix + "c1#f"), m.cout, h, w)
        hmid = sb.raw([g1], rows, m.cout, hw, bias=sd[prefix + "in_layers.2.bias"],
                      bvec=emb_all.data_ptr() + 4 * unet._emb_off[prefix], bvec_ld=unet._emb_total)
        pg.release(a1)
        a2 = sb.post_gn(hmid, None, sd[prefix + "out_layers.0.weight"], sd[prefix + "out_layers.0.bias"], 1e-5, True)
        pg.release(hmid.raw)
        g2, _, _ = conv_args(a2, m.cout, P[prefix + "c2"], P.get(prefix + "c2#f"), m.cout, h, w)
        if m.cin != m.cout:
            c1 = 0 if x1 is None else x1.N
            # the 1x1 skip connection (openaimodel.py:241) rides along as extra K of the second convolution when the plan is a
            # slab-GEMM tile; otherwise it is a GEMM of its own whose slabs land next to the convolution's
            gm, _, _ = conv_args(a2, m.cout, P[prefix + "c2s"], P.get(prefix + "c2s#f"), m.cout, h, w)
            gm.K = 9 * m.cout + m.cin
            gm.skip_a0, gm.skip_c0 = x0.raw.data_ptr(), x0.N
            gm.skip_a1, gm.skip_c1 = (0 if x1 is None else x1.raw.data_ptr()), c1
            cfg, sk = pg.plan(gm, pin)
            if cfg <= 12:                      # no tuned slab plan for the fused shape yet: the plain convolution's plan
                cfg, sk = pg.plan(g2, pin)
                gm.tile_cfg, gm.splitk = cfg, sk
            gm.raw_slabs = 1 if sk > 1 else 0
            gm.splitk_ws, gm.splitk_ws_elems = 1, 1 << 40
            fused = cfg > 12 and pg.lib.ldmk_igemm_check(L.C.byref(gm)) == 0
            gm.splitk_ws, gm.splitk_ws_elems, gm.raw_slabs = 0, 0, 0
            gm._planned = True
            if fused:
                out = sb.raw([gm], rows, m.cout, hw, bias=P[prefix + "c2skip#b"])
            else:
                gs = ops.make_igemm_args(rows, m.cout, m.cin, x0.raw, x0.N, P[prefix + "skip"], None, m.cout, hw,
                                         a1=None if x1 is None else x1.raw, c1=c1, w_frag=P.get(prefix + "skip#f"))
                out = sb.raw([g2, gs], rows, m.cout, hw, bias=P[prefix + "c2skip#b"])
        else:
            assert x1 is None
            out = sb.raw([g2], rows, m.cout, hw, bias=sd[prefix + "out_layers.3.bias"], residual=x0.raw)
        pg.release(a2)
        return out

    def spatial_tf(prefix, m, x, h, w):
        hw, rows = h * w, n * h * w
        C_ = m.heads * m.d_head
        xn = sb.post_gn(x, None, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6, False)
        g = ops.make_igemm_args(rows, C_, m.ch, xn, m.ch, P[prefix + "pin"], None, C_, hw, w_frag=P.get(prefix + "pin#f"))
        hl = sb.raw([g], rows, C_, hw, bias=sd[prefix + "proj_in.bias"])
        pg.release(xn)
        out = None
        for d in range(m.depth):
            q = f"{prefix}transformer_blocks.{d}."
            # --- attn1: LayerNorm materialised by the post launch, QKV slabs read by the attention kernel itself
            hn = sb.post_ln(hl, sd[q + "norm1.weight"], sd[q + "norm1.bias"])
            g = ops.make_igemm_args(rows, 3 * C_, C_, hn, C_, P[q + "qkv"], None, 3 * C_, hw, w_frag=P.get(q + "qkv#uf"))
            sk = pg.plan(g, pin)[1]
            qkv = pg.alloc(sk, rows, 3 * C_)
            pg.igemm_raw(g, qkv)
            pg.release(hn)
            att = pg.alloc(rows, C_)
            pg.add("ldmk_attn_self_small", p_(qkv), sk, rows * 3 * C_, p_(att), n, hw, m.heads, m.d_head ** -0.5)
            pg.release(qkv)
            # --- attn2 with a single context token == a per-sample vector (exact), added with to_out's bias
            v = ctx_pg.alloc(n, C_)
            ctx_pg.add("ldmk_dense_small", p_(ctx_in), unet.context_dim, p_(P[q + "v2"]), 0, p_(v), C_, n, unet.context_dim, C_, ds4)
            cvec = ctx_pg.alloc(n, C_)
            ctx_pg.add("ldmk_dense_small", p_(v), C_, p_(P[q + "o2"]), p_(sd[q + "attn2.to_out.0.bias"]), p_(cvec), C_, n, C_, C_, ds4)
            g = ops.make_igemm_args(rows, C_, C_, att, C_, P[q + "o1"], None, C_, hw, w_frag=P.get(q + "o1#f"))
            h1 = sb.raw([g], rows, C_, hw, bias=sd[q + "attn1.to_out.0.bias"], bvec=cvec.data_ptr(), bvec_ld=C_, residual=hl.raw)
            pg.release(att)
            # --- GEGLU feed-forward
            hn3 = sb.post_ln(h1, sd[q + "norm3.weight"], sd[q + "norm3.bias"])
            f = pg.alloc(rows, 4 * C_)
            g = ops.make_igemm_args(rows, 8 * C_, C_, hn3, C_, P[q + "ff1"], f, 4 * C_, hw, epi=L.EPI_GEGLU, w_frag=P.get(q + "ff1#uf"))
            g.raw_slabs = 1                                        # (planning: GEGLU may split K when the consumer reduces)
            sk = pg.plan(g, pin)[1]
            if sk == 1:
                g.raw_slabs, g.bias = 0, p_(P[q + "ff1b"])
                pg.calls.append((pg.lib.ldmk_igemm, (L.C.byref(g),), g, "ldmk_igemm"))
            else:
                slabs = pg.alloc(sk, rows, 8 * C_)
                pg.igemm_raw(g, slabs)
                pg.post(ops.make_post_args(slabs, rows, 8 * C_, hw, nslab=sk, bias=P[q + "ff1b"], geglu=True, raw_out=f), (slabs, f))
                pg.release(slabs)
            pg.release(hn3)
            if d == m.depth - 1:
                g = ops.make_igemm_args(rows, m.ch, 5 * C_, f, 4 * C_, P[q + "ff2o"], None, m.ch, hw, a1=h1.raw, c1=C_,
                                        w_frag=P.get(q + "ff2o#f"))
                out = sb.raw([g], rows, m.ch, hw, bias=P[q + "ff2o#b"], residual=x.raw)
            else:
                g = ops.make_igemm_args(rows, C_, 4 * C_, f, 4 * C_, P[q + "ff2"], None, C_, hw, w_frag=P.get(q + "ff2#f"))
                hl = sb.raw([g], rows, C_, hw, bias=sd[q + "ff.net.2.bias"], residual=h1.raw)
            pg.release(f)
        return out

    def run_layers(prefix, layers, x0, x1, h, w):
        cur0, cur1 = x0, x1
        for j, m in enumerate(layers):
            p = f"{prefix}{j}."
            if m.kind == "res":
                out = res_block(p, m, cur0, cur1, h, w)
            elif m.kind == "st":
                out = spatial_tf(p, m, cur0, h, w)
            elif m.kind in ("down", "up"):
                xin = sb.plain(cur0)
                up = m.kind == "up"
                g, oh, ow = conv_args(xin, m.ch, P[p + "w"], None, m.ch, h, w, stride=1 if up else 2, upsample=up)
                out = sb.raw([g], n * oh * ow, m.ch, oh * ow, bias=sd[p + ("conv.bias" if up else "op.bias")])
                h, w = oh, ow
            else:
                raise AssertionError(m.kind)
            cur0, cur1 = out, None
        return cur0, h, w

    # ---- the UNet walk (openaimodel.py:729-742)
    hs = []
    h0 = pg.alloc(n * H * W_, mc)
    pg.add("ldmk_conv3x3_in", p_(x_in), cx, p_(cc_in), c_concat, p_(P["input_blocks.0.0.w"]),
           p_(sd["input_blocks.0.0.bias"]), p_(h0), n, H, W_, mc)
    hcur, ch_, cw_ = Lazy(h0, n * H * W_, mc, H * W_), H, W_
    hs.append((hcur, ch_, cw_))
    for i in range(1, len(unet.input_blocks)):
        hcur, ch_, cw_ = run_layers(f"input_blocks.{i}.", unet.input_blocks[i].layers, hcur, None, ch_, cw_)
        hs.append((hcur, ch_, cw_))
    hcur, ch_, cw_ = run_layers("middle_block.", unet.middle_block.layers, hcur, None, ch_, cw_)
    for i, blk in enumerate(unet.output_blocks):
        skip, sh, sw = hs.pop()
        assert (sh, sw) == (ch_, cw_)
        hcur, ch_, cw_ = run_layers(f"output_blocks.{i}.", blk.layers, hcur, skip, ch_, cw_)
    act = sb.post_gn(hcur, None, sd["out.0.weight"], sd["out.0.bias"], 1e-5, True)
    eps = pg.alloc(n, unet.out_channels, H, W_)
    pg.add("ldmk_conv3x3_out_small", p_(act), 0, p_(P["out"]), p_(sd["out.2.bias"]), p_(eps), n, H, W_, unet._final_ch,
           unet.out_channels)
    pg.outputs = dict(eps=eps)
    pg.ctx_program = ctx_pg
    pg.small_route = True
    return pg
