"""MI355X-native UNetModel: drop-in for the YAML `target:` of the reference
(`ldm.modules.diffusionmodules.openaimodel.UNetModel`, openaimodel.py:413-742).

Same constructor kwargs, same parameter tree / state-dict keys, same forward signature
(`forward(x, timesteps, context)`, NCHW fp32 at the boundary).  Underneath, a forward is a
`engine.Program`: a flat list of libldmk.so kernel launches over a static NHWC workspace --
GroupNorm/LayerNorm folded into the consumer GEMM's operand staging, skip-concats as two base
pointers, nearest-upsample folded into the conv gather, the time-embedding and 1-token
cross-attention adds folded into GEMM epilogues.  There is no PyTorch fallback path.
"""
import math
import os

import torch
import torch.nn as nn

from . import lib as L
from . import switches
from . import ops
from .engine import ArithSites, NetBuilder, Program, far_from_tuned, split_enabled as engine_split_enabled, ps_enabled as engine_ps_enabled, f16x2_enabled

# LayerNorm is folded algebraically through the Linear behind it (LDMK_TF_LAYERNORM_FOLDED): exact for well-conditioned rows,
# but it subtracts mean * colsum(W') from x W' in fp32, so rows whose |mean| is many standard deviations lose accuracy
# (measured, one K = 640 GEMM: max error 6e-6 at |mean|/std = 1, 3e-5 near 5, 1.7e-4 at 30, 4.6e-4 at 100).  The statistics
# pass raises a device flag for rows beyond LN_GUARD_RATIO; UNetModel reads it once per program (and once per sampling run) and
# rebuilds itself with the unfolded prologue (LayerNorm applied while the A operand is staged), with a warning.
# LDMK_LN_UNFOLDED=1 starts every model in the unfolded form (A/B switch).
LN_GUARD_RATIO = float(switches.get("LDMK_LN_GUARD_RATIO", "4.0"))


# self attention with K / V pre-split by a pre-pass (csrc/attention_bf16.hip: attn_x3p_fwd_kernel): the pre-pass costs one sweep
# over K and V (and a launch), the key loop saves its K / V splits tokens / 128 times.  Measured in the 64x64x4 step, B = 16
# (profiles/r04_layers64.txt, kernel + pre-pass against ldmk_attn_self_x3): 4096 tokens 1062 -> 994 us per call, 1024 tokens
# 145 -> 152, 256 tokens 28 -> 38: it pays from a few thousand tokens per sample.
ATTN_PRESPLIT_MIN_TOKENS = int(switches.get("LDMK_ATTN_PRESPLIT_MIN_TOKENS", "2048"))
# the F16X2 attention (ldmk_attn_self_h2; it always runs the K / V pre-pass).  Kernel + pre-pass against ldmk_attn_self_x3, B = 16
# (profiles/r04_ab_attn.txt): 4096 tokens 1127 -> 695 us, 1024 tokens 183 -> 122, 256 tokens 38.9 -> 39.4
ATTN_H2_MIN_TOKENS = int(switches.get("LDMK_ATTN_H2_MIN_TOKENS", "512"))


def ln_unfolded_default():
    return bool(switches.get("LDMK_LN_UNFOLDED"))


# ------------------------------------------------------------------------------------------------
# parameter holders: they only own tensors under the reference's names
class _Params(nn.Module):
    def __init__(self, **shapes):
        super().__init__()
        for name, shape in shapes.items():
            self.register_parameter(name, nn.Parameter(torch.empty(*shape), requires_grad=False))


class _Slots(nn.Module):
    """Container whose children are registered under explicit (possibly sparse) integer names, e.g.
    in_layers.0 / in_layers.2 like the nn.Sequential the reference indexes."""

    def __init__(self, **children):
        super().__init__()
        for k, m in children.items():
            self.add_module(k.lstrip("_"), m)


def _conv_params(cin, cout, k):
    return _Params(weight=(cout, cin, k, k), bias=(cout,))


def _lin_params(cin, cout, bias=True):
    return _Params(weight=(cout, cin), bias=(cout,)) if bias else _Params(weight=(cout, cin))


def _norm_params(c):
    return _Params(weight=(c,), bias=(c,))


def _res_block(cin, cout, emb_ch, scale_shift=False, updown=None):
    """ResBlock parameters (openaimodel.py:176-253).  updown = "up" / "down": ResBlock(up=True / down=True) of resblock_updown --
    h_upd / x_upd are parameter-free (:207-216), so the state-dict keys are those of a plain block."""
    ch = dict(in_layers=_Slots(_0=_norm_params(cin), _2=_conv_params(cin, cout, 3)),
              emb_layers=_Slots(_1=_lin_params(emb_ch, 2 * cout if scale_shift else cout)),      # (scale | shift), openaimodel.py:218-224
              out_layers=_Slots(_0=_norm_params(cout), _3=_conv_params(cout, cout, 3)))
    if cin != cout:
        ch["skip_connection"] = _conv_params(cin, cout, 1)
    m = _Slots(**ch)
    m.kind, m.cin, m.cout, m.scale_shift, m.updown = "res", cin, cout, bool(scale_shift), updown
    return m


def _attn(dim, ctx_dim):
    return _Slots(to_q=_lin_params(dim, dim, False), to_k=_lin_params(ctx_dim, dim, False),
                  to_v=_lin_params(ctx_dim, dim, False), to_out=_Slots(_0=_lin_params(dim, dim)))


def _spatial_transformer(ch, heads, d_head, depth, context_dim):
    inner = heads * d_head
    blocks = nn.ModuleList()
    for _ in range(depth):
        blocks.append(_Slots(
            attn1=_attn(inner, inner),
            ff=_Slots(net=_Slots(_0=_Slots(proj=_lin_params(inner, inner * 8)), _2=_lin_params(inner * 4, inner))),
            attn2=_attn(inner, context_dim if context_dim is not None else inner),
            norm1=_norm_params(inner), norm2=_norm_params(inner), norm3=_norm_params(inner)))
    m = _Slots(norm=_norm_params(ch), proj_in=_conv_params(ch, inner, 1), transformer_blocks=blocks,
               proj_out=_conv_params(inner, ch, 1))
    m.kind, m.ch, m.heads, m.d_head, m.depth = "st", ch, heads, d_head, depth
    return m


def _attention_block(ch, heads, new_order=False):
    """AttentionBlock (openaimodel.py:278-324): norm, qkv (Conv1d ch -> 3 ch, kernel 1), proj_out (Conv1d ch -> ch).
    new_order: use_new_attention_order (QKVAttention, :379-407) -- the qkv channels are [q | k | v][head][d] instead of the legacy
    [head][q | k | v][d]."""
    m = _Slots(norm=_norm_params(ch), qkv=_Params(weight=(3 * ch, ch, 1), bias=(3 * ch,)),
               proj_out=_Params(weight=(ch, ch, 1), bias=(ch,)))
    m.kind, m.ch, m.heads, m.d_head, m.new_order = "attn", ch, heads, ch // heads, bool(new_order)
    return m


def _seq(*mods):
    s = nn.Module()
    for i, m in enumerate(mods):
        s.add_module(str(i), m)
    s.layers = list(mods)
    return s


def pack_spatial_transformer(P, sd, prefix, m):
    """Kernel layouts of one SpatialTransformer's weights (attention.py:218-261) into P, keyed by the reference's names."""
    P[prefix + "pin"] = ops.pack_linear(sd[prefix + "proj_in.weight"])
    P[prefix + "pout"] = ops.pack_linear(sd[prefix + "proj_out.weight"])
    for d in range(m.depth):
        q = f"{prefix}transformer_blocks.{d}."
        wqkv = torch.cat([sd[q + "attn1.to_q.weight"], sd[q + "attn1.to_k.weight"], sd[q + "attn1.to_v.weight"]], 0)
        P[q + "qkv"] = ops.pack_linear(wqkv.contiguous())
        P[q + "o1"] = ops.pack_linear(sd[q + "attn1.to_out.0.weight"])
        P[q + "q2"] = ops.pack_linear(sd[q + "attn2.to_q.weight"])
        P[q + "k2"] = ops.pack_linear(sd[q + "attn2.to_k.weight"])
        P[q + "v2"] = ops.pack_linear(sd[q + "attn2.to_v.weight"])
        P[q + "o2"] = ops.pack_linear(sd[q + "attn2.to_out.0.weight"])
        P[q + "ff1"], P[q + "ff1b"] = ops.pack_geglu(sd[q + "ff.net.0.proj.weight"], sd[q + "ff.net.0.proj.bias"])
        P[q + "ff2"] = ops.pack_linear(sd[q + "ff.net.2.weight"])
        # LayerNorm folded through the Linear that follows it (LDMK_TF_LAYERNORM_FOLDED): gamma-scaled weights,
        # their column sums and beta^T W + bias; the unfolded copies above stay for the training step
        for k, nrm, b in (("qkv", "norm1", None), ("q2", "norm2", None), ("ff1", "norm3", P[q + "ff1b"])):
            P[q + k + "_ln"], P[q + k + "_ln#cs"], P[q + k + "_ln#b"] = ops.fold_layernorm(
                P[q + k], sd[q + nrm + ".weight"], sd[q + nrm + ".bias"], b)


def attention_presplit(hw):
    return engine_split_enabled() and hw >= ATTN_PRESPLIT_MIN_TOKENS and switches.get("LDMK_ATTN_PRESPLIT", "1") != "0"


def emit_self_attention(nb_, qkv, att, hw, heads, d_head, att_ps=None, kv_tiles=None):
    """softmax(q k^T / sqrt d) v over token rows [q | k | v] (n hw x 3 C) -> att (n hw x C).  With the split arithmetic on: both
    products fp32-accurate on the 16-bit matrix cores.  From ATTN_H2_MIN_TOKENS tokens per sample, while the model's range flag
    is down (nb_.h2_flag): the F16X2 form -- three fp16 products per term, K / V split once by a pre-pass and moved to LDS by
    LDS-DMA (ldmk_attn_self_h2; 1.47x the bf16x3 kernel at 4096 tokens).  Otherwise bf16x3: from ATTN_PRESPLIT_MIN_TOKENS tokens
    with the K / V pre-pass (ldmk_attn_self_x3p, bitwise ldmk_attn_self_x3; LDMK_ATTN_PRESPLIT=0 turns it off).
    LDMK_SPLIT_BF16=0: the f32 matrix-core kernel."""
    pg, n = nb_.pg, nb_.n
    scale = d_head ** -0.5
    h2_flag = getattr(nb_, "h2_flag", None)
    if d_head != 32:
        # Heads that are not 32 wide (reference kwargs num_heads / num_head_channels: openaimodel.py:443-469,542-549; no shipped
        # YAML sets them -- this path exists so that such checkpoints load and meet the oracle, not for speed): per (sample, head)
        # q k^T and p v as batched GEMMs on head-major copies, the head width zero-padded to a multiple of 32 (exact), scores
        # materialised and normalised by ldmk_softmax_rows -- the VQGAN AttnBlock's route (autoencoder.py).  fp32 throughout.
        assert att_ps is None and kv_tiles is None
        ops, C_ = nb_.ops, heads * d_head
        dp = -(-d_head // 32) * 32
        if hw % 32:
            raise NotImplementedError(f"attention heads of width {d_head} need a token count that is a multiple of 32 (got {hw})")
        q, k, v = (pg.alloc(n * heads, hw, dp) for _ in range(3))
        for i, t in enumerate((q, k, v)):
            pg.add("ldmk_heads_gather", qkv.data_ptr(), 3 * C_, i * C_, t.data_ptr(), n, hw, heads, d_head, dp)
        sc = pg.alloc(n * heads, hw, hw)
        a = ops.make_igemm_args(hw, hw, dp, q, dp, k, sc, hw, hw, b_trans=True, batch=n * heads, a_bstride=hw * dp, w_bstride=hw * dp,
                                out_bstride=hw * hw)
        pg.igemm(a, nb_.pin, per_sample=True)
        pg.add("ldmk_softmax_rows", sc.data_ptr(), n * heads * hw, hw, float(scale))
        o = pg.alloc(n * heads, hw, dp)
        a = ops.make_igemm_args(hw, dp, hw, sc, hw, v, o, dp, hw, batch=n * heads, a_bstride=hw * hw, w_bstride=hw * dp, out_bstride=hw * dp)
        pg.igemm(a, nb_.pin, per_sample=True)
        pg.add("ldmk_heads_scatter", o.data_ptr(), att.data_ptr(), C_, n, hw, heads, d_head, dp)
        nb_.release(q, k, v, sc, o)
        return
    if h2_flag is not None and kv_tiles is not None:
        # the QKV projection's epilogue already wrote K / V as the kernel's pre-split tiles (ldmk_igemm_args.attn_kv_out): no pre-pass
        pg.add("ldmk_attn_self_h2_tiles", qkv.data_ptr(), kv_tiles.data_ptr(), 0 if att is None else att.data_ptr(),
               0 if att_ps is None else att_ps.data_ptr(), h2_flag.data_ptr(), n, hw, heads, scale)
    elif h2_flag is not None and hw >= ATTN_H2_MIN_TOKENS:
        kvs = pg.alloc(pg.lib.ldmk_attn_kv_split_h2_bytes(n, hw, heads), dtype=torch.uint8)
        if att_ps is not None:       # the result in the (F16X2) PS layout only: the A operand of attn1.to_out on a pre-split tile
            pg.add("ldmk_attn_self_h2_ps", qkv.data_ptr(), kvs.data_ptr(), 0 if att is None else att.data_ptr(), att_ps.data_ptr(),
                   h2_flag.data_ptr(), n, hw, heads, scale)
        else:
            pg.add("ldmk_attn_self_h2", qkv.data_ptr(), kvs.data_ptr(), att.data_ptr(), h2_flag.data_ptr(), n, hw, heads, scale)
        nb_.release(kvs)
    elif attention_presplit(hw):
        kvs = pg.alloc(pg.lib.ldmk_attn_kv_split_bytes(n, hw, heads), dtype=torch.uint8)
        if att_ps is not None:       # the result in the PS layout (only): the A operand of attn1.to_out on a pre-split tile
            pg.add("ldmk_attn_self_x3p_ps", qkv.data_ptr(), kvs.data_ptr(), 0 if att is None else att.data_ptr(), att_ps.data_ptr(), n, hw,
                   heads, scale)
        else:
            pg.add("ldmk_attn_self_x3p", qkv.data_ptr(), kvs.data_ptr(), att.data_ptr(), n, hw, heads, scale)
        nb_.release(kvs)
    else:
        assert att_ps is None
        pg.add("ldmk_attn_self_x3" if engine_split_enabled() else "ldmk_attn_self", qkv.data_ptr(), att.data_ptr(), n, hw, heads, scale)


def pack_attention_block(P, sd, prefix, m):
    """AttentionBlock weights in kernel layouts.  The reference's qkv Conv1d orders its 3 ch output channels
    [head][q | k | v][32] (QKVAttentionLegacy splits the heads first, openaimodel.py:366-367); the attention kernels read token
    rows [q of every head | k of every head | v of every head], so the output channels (and the bias) are permuted here."""
    ch, heads = m.ch, m.heads
    d = ch // heads
    if getattr(m, "new_order", False):     # QKVAttention: [q | k | v][head][d] is the kernels' own order
        w, bq = sd[prefix + "qkv.weight"].reshape(3 * ch, ch).contiguous(), sd[prefix + "qkv.bias"].contiguous()
    else:
        w = sd[prefix + "qkv.weight"].reshape(heads, 3, d, ch).permute(1, 0, 2, 3).reshape(3 * ch, ch).contiguous()
        bq = sd[prefix + "qkv.bias"].reshape(heads, 3, d).permute(1, 0, 2).reshape(3 * ch).contiguous()
    P[prefix + "aqkv"] = ops.pack_linear(w)
    P[prefix + "aqkv#b"] = bq
    P[prefix + "apout"] = ops.pack_linear(sd[prefix + "proj_out.weight"].reshape(ch, ch).contiguous())


def emit_attention_block(nb_, P, sd, prefix, m, x, h, w):
    """AttentionBlock._forward (openaimodel.py:316-324) as launches into nb_.pg: GroupNorm32 (eps 1e-5, no activation) folded
    into the qkv projection's A staging, flash self-attention (q and k each scaled by d^-1/4 in the reference = the logits by
    d^-1/2, which is the kernels' pre-scale of Q), proj_out + the block input + the GroupNorm records of the result.
    x: NHWC (n, h, w, C) -> NHWC."""
    pg, n = nb_.pg, nb_.n
    p_ = lambda t: 0 if t is None else t.data_ptr()
    hw = h * w
    rows = n * hw
    xr = x.reshape(rows, m.ch)
    coef = nb_.gn(x, None, hw, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-5)
    with nb_.site(prefix + "attn"):
        qkv = nb_.lin(xr, P[prefix + "aqkv"], P[prefix + "aqkv#b"], hw, tf=L.TF_AFFINE, tf_coef=coef, wf=P.get(prefix + "aqkv#f"))
        nb_.release(coef)
        att = pg.alloc(rows, m.ch)
        emit_self_attention(nb_, qkv, att, hw, m.heads, m.d_head)
        nb_.release(qkv)
    with nb_.site(prefix + "proj_out"):
        out = nb_.lin(att, P[prefix + "apout"], sd[prefix + "proj_out.bias"], hw, residual=xr, stats=True, wf=P.get(prefix + "apout#f"))
    nb_.release(att)
    return out.view(n, h, w, m.ch)


def pack_gemm_copies(P, unfolded=False):
    """The second copies of the GEMM weights the plans may ask for, keyed `<weight key>#f` / `#s`."""
    # fragment-order copies of the token-row Linear weights: the row GEMM (csrc/rgemm.hip) reads these
    # (and the slab GEMM of the small-batch route, csrc/sgemm.hip, which also takes the ResBlock convolutions: c1 / c2)
    for k in [k for k in P if k.rsplit(".", 1)[-1] in ("pin", "pout", "qkv_ln", "o1", "q2_ln", "o2", "ff1_ln", "ff2", "skip",
                                                        "c1", "c2", "aqkv", "apout")
              or (unfolded and k.rsplit(".", 1)[-1] in ("qkv", "ff1", "q2"))]:
        wf = ops.pack_wfrag(P[k])
        if wf is not None:
            P[k + "#f"] = wf
    # bf16x3 images of the GEMM weights (LDMK_COMPUTE_BF16X3, include/ldmk.h): fp32-accurate products at the bf16 matrix
    # rate; engine.Program.plan() uses them for the shapes dsml_thesis_amd/igemm_plans_x3.json lists
    if engine_split_enabled():
        for k in list(P):
            tail = k.rsplit(".", 1)[-1]
            if tail in ("pin", "pout", "qkv_ln", "o1", "ff1_ln", "ff2", "skip", "c1", "c2", "w", "aqkv", "apout") or (unfolded and tail in ("qkv", "ff1")):
                if P[k].dim() == 2:
                    P[k + "#s"] = ops.pack_wsplit(P[k])
                    if f16x2_enabled():          # ... and the two fp16 images of the F16X2 arithmetic
                        P[k + "#h"] = ops.pack_wsplit_h2(P[k])
            elif tail in ("c1#wg", "c2#wg", "w#up"):
                P[k + "#s"] = ops.pack_wsplit(P[k], batch=P[k].shape[0])
                if f16x2_enabled():
                    P[k + "#h"] = ops.pack_wsplit_h2(P[k], batch=P[k].shape[0])
    # PS-layout copies (csrc/igemm_ps.hip: both operands pre-split, moved to LDS by LDS-DMA) for the GEMMs whose A operand a
    # producer can write in that layout: LayerNorm-folded projections (the statistics pass writes it), ff.net.2 (the GEGLU epilogue)
    if engine_ps_enabled():
        for k in list(P):
            tail = k.rsplit(".", 1)[-1]
            if tail in ("qkv_ln", "ff1_ln", "ff2", "o1", "pout") and P[k].dim() == 2 and P[k].shape[0] % 32 == 0 and P[k].shape[1] % 32 == 0:
                P[k + "#p"] = ops.pack_wps(P[k])
            elif tail in ("c1#wg", "c2#wg", "w#up") and P[k].shape[1] % 32 == 0 and P[k].shape[2] % 32 == 0:
                P[k + "#p"] = ops.pack_wps(P[k], batch=P[k].shape[0])     # Winograd planes / upsampling phases (transforms write V in PS)
        # ... and in the two-plane fp16 form of the F16X2 arithmetic (`#p2`; a program takes the form of its arithmetic)
        if f16x2_enabled():
            for k in list(P):
                tail = k.rsplit(".", 1)[-1]
                if tail in ("qkv_ln", "ff1_ln", "ff2", "o1", "pout") and P[k].dim() == 2 and P[k].shape[0] % 32 == 0 and P[k].shape[1] % 32 == 0:
                    P[k + "#p2"] = ops.pack_wps(P[k], h2=True)
                elif tail in ("c1#wg", "c2#wg", "w#up") and P[k].shape[1] % 32 == 0 and P[k].shape[2] % 32 == 0:
                    P[k + "#p2"] = ops.pack_wps(P[k], batch=P[k].shape[0], h2=True)
                elif tail in ("c1", "c2") and P[k].dim() == 2 and P[k].shape[0] % (9 * 32) == 0 and P[k].shape[1] % 32 == 0:
                    # the ResBlock convolutions for the conv-mode pre-split tile (engine.NetBuilder.ps_query_conv): pack_conv3x3's
                    # [9 C_in][C_out] matrix is already in the tile's K order (32-channel chunk major, tap minor)
                    P[k + "#pc2"] = ops.pack_wps(P[k], h2=True)


def emit_spatial_transformer(nb_, ctx_pg, P, sd, prefix, m, x, h, w, L_ctx, ctx_in, context_dim, unfolded=False, ln_flag=None):
    """SpatialTransformer.forward (attention.py:250-261) as launches into nb_.pg: GroupNorm folded into proj_in, per block
    LN1 -> fused QKV -> flash self-attention -> to_out (+ residual, + the 1-token cross-attention vector), [LN2 -> to_q ->
    cross-attention -> to_out for longer contexts], LN3 -> GEGLU projection -> ff.net.2 (+ residual), then proj_out (+ the
    block input, + the GroupNorm records of the result).  Context-only projections go to `ctx_pg` (run once per sample() call).
    x: NHWC (n, h, w, C) -> NHWC.  `unfolded`: LayerNorm in the A staging instead of folded through the product; `ln_flag`:
    device int the folded form's statistics passes raise for mean-dominated rows (LN_GUARD_RATIO)."""
    pg, n = nb_.pg, nb_.n
    gn, lin = nb_.gn, nb_.lin
    p_ = lambda t: 0 if t is None else t.data_ptr()
    hw = h * w
    C_ = m.heads * m.d_head
    rows = n * hw
    xr = x.reshape(rows, m.ch)
    coef = gn(x, None, hw, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6)
    with nb_.site(prefix + "proj_in"):
        hcur = lin(xr, P[prefix + "pin"], sd[prefix + "proj_in.bias"], hw, tf=L.TF_AFFINE, tf_coef=coef, wf=P.get(prefix + "pin#f"))
    nb_.release(coef)
    stats = pg.alloc(rows, 2)

    # The arithmetic is a property of the SITE (engine.ArithSites): `attn1` = LN1 -> fused QKV -> self attention -> to_out (they hand
    # each other split operands: the QKV epilogue writes the attention's K / V tiles, the attention writes to_out's A operand), `ff` =
    # LN3 -> GEGLU projection -> ff.net.2 [-> proj_out on a pre-split tile], `attn2` (contexts of several tokens), `proj_in`,
    # `proj_out`.  Inside `with nb_.site(...)` nb_.h2_flag is that site's range flag, or None when the site runs in bf16x3.
    hf = lambda: getattr(nb_, "h2_flag", None)
    psfx = lambda: "#p2" if hf() is not None else "#p"       # the PS weight copies in the form of the site's arithmetic

    kv_state = {"tiles": None}

    def ln_lin(x2d, wkey, geglu, out_ps=None, attn_kv=None):
        """LayerNorm statistics + the Linear the LayerNorm is folded through.  With a pre-split plan for the GEMM
        (engine.ps_query: csrc/igemm_ps.hip) the statistics pass also writes the rows in the PS layout and the GEMM moves them
        to LDS by LDS-DMA -- every element is split once, by this pass, instead of once per N-tile inside the GEMM.
        `out_ps`: (GEGLU only) the result goes out in the PS layout ONLY, for ff.net.2."""
        wp = P[wkey]
        N_ = wp.shape[1]
        h2_flag = hf()
        plan = nb_.ps_query(rows, N_, C_, tf=L.TF_LAYERNORM_FOLDED, epi=L.EPI_GEGLU if geglu else L.EPI_NONE) if wkey + psfx() in P else None
        if plan is not None:
            xs = pg.alloc_ps(rows, C_)
            if h2_flag is not None:
                pg.add("ldmk_ln_stats_ps_h2", p_(x2d), rows, C_, 1e-5, p_(stats), p_(xs), LN_GUARD_RATIO, p_(ln_flag), p_(h2_flag))
            else:
                pg.add("ldmk_ln_stats_ps", p_(x2d), rows, C_, 1e-5, p_(stats), p_(xs), LN_GUARD_RATIO, p_(ln_flag))
            kw = {}
            if attn_kv is not None and h2_flag is not None and int(plan[0]) in (23, 27) and int(plan[1]) <= 1:
                # the fused QKV projection writes K / V straight as the attention's pre-split tiles (no fp32 K / V, no pre-pass)
                kw["attn_kv"] = attn_kv
                kv_state["tiles"] = attn_kv[0]
            y = nb_.lin_ps(plan, rows, C_, xs, wp, P[wkey + psfx()], P[wkey + "#b"], hw, geglu=geglu, out_ps=out_ps,
                           tf=L.TF_LAYERNORM_FOLDED, row_stats=stats, ln_colsum=P[wkey + "#cs"], **kw)
            nb_.release(xs)
            return y
        assert out_ps is None
        pg.add("ldmk_ln_stats_guard", p_(x2d), rows, C_, 1e-5, p_(stats), LN_GUARD_RATIO, p_(ln_flag))
        return lin(x2d, wp, P[wkey + "#b"], hw, geglu=geglu, tf=L.TF_LAYERNORM_FOLDED, row_stats=stats,
                   ln_colsum=P[wkey + "#cs"], wf=P.get(wkey + "#f"))

    hc_ps = plan_p = None
    ff_site = None

    def attn1_section(q, hcur):
        """LN1 folded into the fused QKV GEMM, flash attention, to_out + residual [+ the 1-token cross-attention vector]: one site."""
        h2_flag = hf()
        if unfolded:
            pg.add("ldmk_ln_stats", p_(hcur), rows, C_, 1e-5, p_(stats))
            qkv = lin(hcur, P[q + "qkv"], None, hw, tf=L.TF_LAYERNORM, row_stats=stats,
                      ln_gamma=sd[q + "norm1.weight"], ln_beta=sd[q + "norm1.bias"], wf=P.get(q + "qkv#f"))
        else:
            kv_state["tiles"] = None
            kvt = None
            if (h2_flag is not None and hw >= ATTN_H2_MIN_TOKENS and hw % 64 == 0 and m.d_head == 32
                    and switches.get("LDMK_QKV_TILES", "1") != "0"):
                kvt = pg.alloc(pg.lib.ldmk_attn_kv_split_h2_bytes(n, hw, m.heads), dtype=torch.uint8)
            qkv = ln_lin(hcur, q + "qkv_ln", False, attn_kv=None if kvt is None else (kvt, hw, m.heads))
            if kvt is not None and kv_state["tiles"] is None:
                nb_.release(kvt)            # (no pre-split plan for this projection: the attention runs its own pre-pass)
                kvt = None
        # both attention products in the fp32-accurate bf16x3 arithmetic (24 bf16 MFMAs of 32 cycles per 32 keys
        # instead of 32 fp32 ones of 64); LDMK_SPLIT_BF16=0 keeps the f32 matrix-core kernel.  With a pre-split plan for
        # attn1.to_out the attention kernel writes its result in the PS layout only (from its accumulators, no LDS pass)
        plan_o = (nb_.ps_query(rows, C_, C_) if (q + "o1" + psfx() in P and L_ctx == 1 and hw % 32 == 0 and switches.get("LDMK_ATTN_PS", "1") != "0"
                                                 and m.d_head == 32
                                                 and (hw >= ATTN_H2_MIN_TOKENS if h2_flag is not None else attention_presplit(hw))) else None)
        att = None if plan_o is not None else pg.alloc(rows, C_)
        att_ps = pg.alloc_ps(rows, C_) if plan_o is not None else None
        kv_tiles = None if unfolded else kv_state["tiles"]
        emit_self_attention(nb_, qkv, att, hw, m.heads, m.d_head, att_ps=att_ps, kv_tiles=kv_tiles)
        nb_.release(qkv)
        if kv_tiles is not None:
            nb_.release(kv_tiles)
        if L_ctx != 1:
            return lin(att, P[q + "o1"], sd[q + "attn1.to_out.0.bias"], hw, residual=hcur, out=hcur, wf=P.get(q + "o1#f")), att
        # attn2 with a single context token: softmax over one key == 1, so the block adds to_out(to_v(ctx)) to every
        # position (exact); to_q/norm2 are dead.  The vector rides in to_out's epilogue.
        v = ctx_pg.alloc(n, C_)
        ctx_pg.add("ldmk_dense_small", p_(ctx_in), context_dim, p_(P[q + "v2"]), 0, p_(v), C_, n,
                   context_dim, C_, 0)
        cvec = ctx_pg.alloc(n, C_)
        ctx_pg.add("ldmk_dense_small", p_(v), C_, p_(P[q + "o2"]), p_(sd[q + "attn2.to_out.0.bias"]), p_(cvec),
                   C_, n, C_, C_, 0)
        if plan_o is not None:
            h1 = nb_.lin_ps(plan_o, rows, C_, att_ps, P[q + "o1"], P[q + "o1" + psfx()], sd[q + "attn1.to_out.0.bias"], hw, out=hcur,
                            residual=hcur, batch_vec=cvec, batch_vec_ld=C_)
            nb_.release(att_ps)
        else:
            h1 = lin(att, P[q + "o1"], sd[q + "attn1.to_out.0.bias"], hw, residual=hcur, out=hcur,
                     batch_vec=cvec, batch_vec_ld=C_, wf=P.get(q + "o1#f"))   # + the per-sample cross-attention vector
            nb_.release(att)
        return h1, None

    def attn2_section(q, h1, att):
        """Cross attention over a context of several tokens: K / V projections in the context program, LN2 -> to_q -> attention -> to_out."""
        kk = ctx_pg.alloc(n * L_ctx, C_)
        vv = ctx_pg.alloc(n * L_ctx, C_)
        ctx_pg.add("ldmk_dense_small", p_(ctx_in), context_dim, p_(P[q + "k2"]), 0, p_(kk), C_,
                   n * L_ctx, context_dim, C_, 0)
        ctx_pg.add("ldmk_dense_small", p_(ctx_in), context_dim, p_(P[q + "v2"]), 0, p_(vv), C_,
                   n * L_ctx, context_dim, C_, 0)
        if unfolded:
            pg.add("ldmk_ln_stats", p_(h1), rows, C_, 1e-5, p_(stats))
            q2 = lin(h1, P[q + "q2"], None, hw, tf=L.TF_LAYERNORM, row_stats=stats, ln_gamma=sd[q + "norm2.weight"],
                     ln_beta=sd[q + "norm2.bias"], out=att, wf=P.get(q + "q2#f"))
        else:
            pg.add("ldmk_ln_stats_guard", p_(h1), rows, C_, 1e-5, p_(stats), LN_GUARD_RATIO, p_(ln_flag))
            q2 = lin(h1, P[q + "q2_ln"], P[q + "q2_ln#b"], hw, tf=L.TF_LAYERNORM_FOLDED, row_stats=stats,
                     ln_colsum=P[q + "q2_ln#cs"], out=att, wf=P.get(q + "q2_ln#f"))
        a2 = pg.alloc(rows, C_)
        if m.d_head == 32:
            pg.add("ldmk_attn_cross", p_(q2), C_, p_(kk), p_(vv), C_, p_(a2), C_, n, hw, L_ctx, m.heads, m.d_head ** -0.5)
        elif m.d_head in (40, 64, 80):
            pg.add("ldmk_attn_cross_d", p_(q2), C_, p_(kk), p_(vv), C_, p_(a2), C_, n, hw, L_ctx, m.heads, m.d_head, m.d_head ** -0.5)
        else:
            raise NotImplementedError(f"cross attention over a context of {L_ctx} tokens with heads of width {m.d_head} (built: 32, 40, 64, 80)")
        h2 = lin(a2, P[q + "o2"], sd[q + "attn2.to_out.0.bias"], hw, residual=h1, out=h1, wf=P.get(q + "o2#f"))
        nb_.release(att, a2)
        return h2

    def ff_section(q, h2, last):
        """GEGLU feed-forward: LN3 folded into the first GEMM, gate in its epilogue, ff.net.2 + residual.  When both GEMMs run on
        pre-split tiles (engine.ps_query) the intermediate exists in the PS layout only -- written by the GEGLU epilogue, split
        once -- never as an fp32 tensor.  Returns (hcur, hc_ps, plan_p)."""
        Nf = P[q + "ff2"].shape[0]
        plan_g = plan_f = None
        if not unfolded and q + "ff1_ln" + psfx() in P and q + "ff2" + psfx() in P:
            plan_g = nb_.ps_query(rows, 2 * Nf, C_, tf=L.TF_LAYERNORM_FOLDED, epi=L.EPI_GEGLU)
            plan_f = nb_.ps_query(rows, C_, Nf)
        if plan_g is not None and plan_f is not None:
            f_ps = pg.alloc_ps(rows, Nf)
            ln_lin(h2, q + "ff1_ln", True, out_ps=f_ps)
            # the last block's ff.net.2 also writes its result pre-split when proj_out runs on a pre-split tile
            # (proj_out on a pre-split tile: built and tested, off by default -- its epilogue carries the GroupNorm records of the block
            #  output, the lane = column form of the kernel, and loses to the row GEMM in the step: 951.5 vs 959.6 sample-steps/s with
            #  attn1.to_out off as well, A/B on one box; LDMK_POUT_PS=1 turns it on.  A split-K ff.net.2 has no PS epilogue.)
            plan_p = (nb_.ps_query(rows, m.ch, C_) if (last and prefix + "pout" + psfx() in P and plan_f[1] <= 1
                                                       and switches.get("LDMK_POUT_PS", "0") == "1") else None)
            hc_ps = pg.alloc_ps(rows, C_) if plan_p is not None else None
            hcur = nb_.lin_ps(plan_f, rows, Nf, f_ps, P[q + "ff2"], P[q + "ff2" + psfx()], sd[q + "ff.net.2.bias"], hw, out=h2, residual=h2,
                              out_ps=hc_ps)
            nb_.release(f_ps)
            return hcur, hc_ps, plan_p
        if unfolded:
            pg.add("ldmk_ln_stats", p_(h2), rows, C_, 1e-5, p_(stats))
            f = lin(h2, P[q + "ff1"], P[q + "ff1b"], hw, geglu=True, tf=L.TF_LAYERNORM, row_stats=stats,
                    ln_gamma=sd[q + "norm3.weight"], ln_beta=sd[q + "norm3.bias"], wf=P.get(q + "ff1#f"))
        else:
            f = ln_lin(h2, q + "ff1_ln", True)
        hcur = lin(f, P[q + "ff2"], sd[q + "ff.net.2.bias"], hw, residual=h2, out=h2, wf=P.get(q + "ff2#f"))
        nb_.release(f)
        return hcur, None, None

    for d in range(m.depth):
        q = f"{prefix}transformer_blocks.{d}."
        with nb_.site(q + "attn1"):
            h1, att = attn1_section(q, hcur)
        if L_ctx == 1:
            h2 = h1
        else:
            with nb_.site(q + "attn2"):
                h2 = attn2_section(q, h1, att)
        ff_site = q + "ff"
        with nb_.site(ff_site):
            hcur, hc_ps, plan_p = ff_section(q, h2, d == m.depth - 1)
    if hc_ps is not None:
        with nb_.site(ff_site):       # (proj_out reads the operand ff.net.2 wrote pre-split: same site, same arithmetic)
            out = nb_.lin_ps(plan_p, rows, C_, hc_ps, P[prefix + "pout"], P[prefix + "pout" + psfx()], sd[prefix + "proj_out.bias"], hw, residual=xr,
                             stats=True)
        nb_.release(hc_ps)
    else:
        with nb_.site(prefix + "proj_out"):
            out = lin(hcur, P[prefix + "pout"], sd[prefix + "proj_out.bias"], hw, residual=xr, stats=True, wf=P.get(prefix + "pout#f"))
    nb_.release(hcur, stats)
    return out.view(n, h, w, m.ch)


# A sampling run is evaluated again while the model changes its arithmetic underneath it: at most this many evaluations
# (F16X2 sites denied once or twice, then the whole model in bf16x3, then the unfolded LayerNorm: UNetModel.flags_tripped)
MAX_ARITHMETIC_PASSES = 5


def rerun_if_flags_tripped(unet_of):
    """Decorator for a sampling loop (a method whose object leads to the UNetModel through `unet_of(self)`).  The loop itself never
    reads a device flag (it may be a hipGraph replay); after the run ONE host read of the model's flags
    (UNetModel.flags_tripped: F16X2 range flags per site, the folded-LayerNorm guard).  If any was up, the model has re-planned
    the sites concerned and the run is repeated -- the SAME run: the CUDA generator is put back to where the first pass found it,
    so start noise and per-step noise drawn inside the loop come out again (graph capture is generator-neutral,
    engine.GraphedProgram), and the result is the one a model started in the final arithmetic gives for the same seed.
    `callback` / `img_callback` fire during every pass, a discarded one included: side effects there must tolerate a repeat."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(self, *args, **kwargs):
            if torch.cuda.is_current_stream_capturing():
                return fn(self, *args, **kwargs)
            unet = unet_of(self)
            rng = torch.cuda.get_rng_state()
            for _ in range(MAX_ARITHMETIC_PASSES):
                out = fn(self, *args, **kwargs)
                if not unet.flags_tripped():
                    break
                torch.cuda.set_rng_state(rng)
            return out
        return wrapped
    return deco


rerun_if_layernorm_guard_tripped = rerun_if_flags_tripped      # (the name rounds 2-4 used)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        # combinations the reference accepts but this path does not implement fail loudly
        if dims != 2:
            raise NotImplementedError("UNetModel: only dims=2")
        if n_embed is not None:
            raise NotImplementedError("UNetModel: n_embed (the id-predictor head) is not part of the sampling path built here")
        if use_new_attention_order and use_spatial_transformer:
            pass        # (QKVAttention lives in AttentionBlock only: the flag is inert with spatial transformers, openaimodel.py:379,557-570)
        if not conv_resample:
            raise NotImplementedError("UNetModel: conv_resample=False")
        if use_fp16:
            raise NotImplementedError("UNetModel: fp32 only (reference precision)")
        if dropout != 0 and self.training:
            pass  # inference path: dropout is the identity
        if use_spatial_transformer:
            assert context_dim is not None, "use_spatial_transformer needs context_dim (openaimodel.py:471-472)"
        else:
            assert context_dim is None, "context_dim needs use_spatial_transformer (openaimodel.py:474-475)"
        self.use_spatial_transformer = bool(use_spatial_transformer)
        if isinstance(context_dim, (list, tuple)) or type(context_dim).__name__ == "ListConfig":
            context_dim = list(context_dim)
            assert len(context_dim) == 1, "a single context dim is supported"
            context_dim = context_dim[0]
        if num_heads == -1:
            assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        attention_resolutions = list(attention_resolutions)
        channel_mult = list(channel_mult)
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_res_blocks = out_channels, num_res_blocks
        self.attention_resolutions, self.channel_mult = attention_resolutions, channel_mult
        self.dropout, self.conv_resample, self.num_classes = dropout, conv_resample, num_classes
        self.use_checkpoint, self.dtype = use_checkpoint, torch.float32
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.context_dim, self.transformer_depth = context_dim, transformer_depth
        self.use_scale_shift_norm, self.use_new_attention_order = bool(use_scale_shift_norm), bool(use_new_attention_order)
        self.resblock_updown = bool(resblock_updown)
        mc = model_channels
        emb_ch = 4 * mc
        self._heads32 = True
        self.time_embed = _Slots(_0=_lin_params(mc, emb_ch), _2=_lin_params(emb_ch, emb_ch))
        if num_classes is not None:          # class-conditional ('adm') UNets: emb += label_emb(y), openaimodel.py:513-514,726-728
            self.label_emb = _Params(weight=(int(num_classes), emb_ch))

        def heads_for(ch):
            if num_head_channels == -1:
                n = num_heads
            else:
                n = ch // num_head_channels
            d = ch // n if legacy else (num_head_channels if num_head_channels != -1 else ch // n)
            if n * d != ch or d % 4:
                raise NotImplementedError(f"UNetModel: {ch} channels do not split into {n} heads of a width that is a multiple of 4")
            self._heads32 = self._heads32 and d == 32      # the flash kernels are built for 32; other widths: batched GEMMs
            return n, d

        def st(ch):
            if not use_spatial_transformer:
                # AttentionBlock(ch, num_heads, num_head_channels = dim_head): heads = ch // num_head_channels when that is set,
                # else num_heads (openaimodel.py:296-303,545-556; legacy: dim_head = num_head_channels)
                n = num_heads if num_head_channels == -1 else ch // num_head_channels
                if n * (ch // n) != ch or (ch // n) % 4:
                    raise NotImplementedError(f"UNetModel: {ch} channels do not split into {n} heads of a width that is a multiple of 4")
                self._heads32 = self._heads32 and ch // n == 32
                return _attention_block(ch, n, use_new_attention_order)
            n, d = heads_for(ch)
            return _spatial_transformer(ch, n, d, transformer_depth, context_dim)

        # ---- same walk as openaimodel.py:513-681
        first = _conv_params(in_channels, mc, 3)
        first.kind = "conv_in"
        self.input_blocks = nn.ModuleList([_seq(first)])
        chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [_res_block(ch, mult * mc, emb_ch, use_scale_shift_norm)]
                ch = mult * mc
                if ds in attention_resolutions:
                    layers.append(st(ch))
                self.input_blocks.append(_seq(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                if resblock_updown:              # ResBlock(ch, ..., out_channels=ch, down=True), openaimodel.py:570-584
                    down = _res_block(ch, ch, emb_ch, use_scale_shift_norm, updown="down")
                else:
                    down = _Slots(op=_conv_params(ch, ch, 3))
                    down.kind, down.ch = "down", ch
                self.input_blocks.append(_seq(down))
                chans.append(ch)
                ds *= 2
        self.middle_block = _seq(_res_block(ch, ch, emb_ch, use_scale_shift_norm), st(ch), _res_block(ch, ch, emb_ch, use_scale_shift_norm))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [_res_block(ch + ich, mc * mult, emb_ch, use_scale_shift_norm)]
                ch = mc * mult
                if ds in attention_resolutions:
                    layers.append(st(ch))
                if level and i == num_res_blocks:
                    if resblock_updown:          # ResBlock(ch, ..., out_channels=ch, up=True), openaimodel.py:660-674
                        up = _res_block(ch, ch, emb_ch, use_scale_shift_norm, updown="up")
                    else:
                        up = _Slots(conv=_conv_params(ch, ch, 3))
                        up.kind, up.ch = "up", ch
                    layers.append(up)
                    ds //= 2
                self.output_blocks.append(_seq(*layers))
        self.out = _Slots(_0=_norm_params(ch), _2=_conv_params(mc, out_channels, 3))
        self._final_ch = ch
        self.reset_parameters()
        self._packed = None
        self._pack_sig = None
        self._programs = {}
        self._ctx_sig = None
        self.auto_repack = True
        # LayerNorm folded through the product unless the guard (LN_GUARD_RATIO) has found mean-dominated rows in this model
        self.ln_unfolded = ln_unfolded_default()
        self._ln_flag = None
        # the F16X2 arithmetic where a kernel offers it; a SITE whose range flag goes up is denied it (engine.ArithSites), the
        # whole model only after H2_SITE_PASSES rounds of that in a row
        self.f16x2 = f16x2_enabled()
        self._sites = None
        self._denied0 = set()               # (deny_f16x2 before the first pack)
        self._h2_passes = 0
        self.generation = 0                 # bumped whenever the launch programs are dropped (samplers key their caches on it)
        # tile shapes are chosen from the problem size; set policy_batch = G to choose them as if the batch
        # were G, which makes per-sample results bitwise identical however a G-sample job is sharded
        self.policy_batch = None

    # ---- initialisation: PyTorch layer defaults + the reference's zero_module sites ----------------
    @torch.no_grad()
    def reset_parameters(self):
        zero = set()
        for name, _ in self.named_parameters():
            if ".out_layers.3." in name or ".proj_out." in name or name.startswith("out.2."):
                zero.add(name)      # openaimodel.py:229-231,685 ; attention.py:244-248
        for name, p in self.named_parameters():
            if name in zero:
                p.zero_()
            elif p.dim() >= 2:
                fan_in = p[0].numel()
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            elif name.endswith(".bias"):
                owner = dict(self.named_parameters()).get(name[:-4] + "weight")
                if owner is not None and owner.dim() >= 2:
                    fan_in = owner[0].numel()
                    p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
                else:
                    p.zero_()       # norm bias
            else:
                p.fill_(1.0)        # norm weight

    # ---- weight packing (after load_state_dict / EMA swap): torch layouts -> kernel layouts ---------
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    @torch.no_grad()
    def pack_weights(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.LdmkError("UNetModel: parameters must live on a GPU (model.cuda()); there is no CPU path")
        P = {}
        sd = {k: v.detach().float().contiguous() for k, v in self.named_parameters()}

        def res(prefix, m):
            P[prefix + "c1"] = ops.pack_conv3x3(sd[prefix + "in_layers.2.weight"])
            P[prefix + "c2"] = ops.pack_conv3x3(sd[prefix + "out_layers.3.weight"])
            # Winograd-domain weights for the convolutions wide enough to take that route (NetBuilder.gn_conv decides per size)
            if m.cin >= NetBuilder.WINO_MIN_CIN:
                P[prefix + "c1#wg"] = ops.pack_winograd(sd[prefix + "in_layers.2.weight"])
            if m.cout >= NetBuilder.WINO_MIN_CIN:
                P[prefix + "c2#wg"] = ops.pack_winograd(sd[prefix + "out_layers.3.weight"])
            if m.cin != m.cout:
                P[prefix + "skip"] = ops.pack_linear(sd[prefix + "skip_connection.weight"])

        def stp(prefix, m):
            pack_spatial_transformer(P, sd, prefix, m)

        emb_w, emb_b = [], []
        self._emb_off = {}
        off = 0
        for prefix, m in self._walk():
            if m.kind == "res":
                res(prefix, m)
                emb_w.append(sd[prefix + "emb_layers.1.weight"])
                emb_b.append(sd[prefix + "emb_layers.1.bias"])
                self._emb_off[prefix] = off
                off += 2 * m.cout if m.scale_shift else m.cout
            elif m.kind == "st":
                stp(prefix, m)
            elif m.kind == "attn":
                pack_attention_block(P, sd, prefix, m)
            elif m.kind == "conv_in":
                P[prefix + "w"] = ops.pack_conv3x3_narrow(sd[prefix + "weight"])
            elif m.kind == "down":
                P[prefix + "w"] = ops.pack_conv3x3(sd[prefix + "op.weight"])
            elif m.kind == "up":
                P[prefix + "w"] = ops.pack_conv3x3(sd[prefix + "conv.weight"])
                if sd[prefix + "conv.weight"].shape[1] >= NetBuilder.WINO_MIN_CIN:      # four 2x2-tap phase convolutions
                    P[prefix + "w#up"] = ops.pack_upconv(sd[prefix + "conv.weight"])
        # every ResBlock's emb_layers Linear batched into one [4mc][sum cout] matrix
        P["emb_all"] = ops.pack_linear(torch.cat(emb_w, 0).contiguous())
        P["emb_all_b"] = torch.cat(emb_b, 0).contiguous()
        self._emb_total = off
        P["te0"] = ops.pack_linear(sd["time_embed.0.weight"])
        P["te2"] = ops.pack_linear(sd["time_embed.2.weight"])
        P["out"] = ops.pack_conv3x3_narrow(sd["out.2.weight"])
        P["freqs"] = ops.timestep_freqs(self.model_channels, device=dev)
        pack_gemm_copies(P, self.ln_unfolded)
        self._ln_flag = torch.zeros(1, device=dev, dtype=torch.int32)
        if self._sites is None or self._sites.flags.device != dev:
            denied = set(self._denied0) if self._sites is None else self._sites.denied
            self._sites = ArithSites(dev)          # (the denied set survives a re-pack: EMA swaps change the weights a little,
            self._sites.denied = denied            #  not the layers whose operands are large)
        self.generation += 1
        self._sd = sd
        self._packed = P
        self._pack_sig = self._signature()
        self._programs = {}
        self._ctx_sig = None
        self._weight_token = None           # (adopt_weights names the set again right after its own call)

    # ---- packed-weight sets by name: the EMA swap without the re-pack -------------------------------------------
    def adopt_weights(self, token):
        """Tell the model WHICH weights its parameters hold right now (any hashable token: LatentDiffusion.ema_scope passes
        ('ema', version of the shadow buffers) on entry and ('train', version of the stored weights) on exit).  The first time a
        token is seen the kernel-layout copies are packed from the parameters and kept under it -- with their launch programs and
        captured graphs; the next time they are simply put back, so `with model.ema_scope(): sample(...)` pays for the ~18 bytes per
        parameter of weight forms, the program builds and the graph captures ONCE, not on every entry and every exit.  At most two
        sets are kept (the one in use and the previous one).  The caller vouches that equal tokens mean equal values; parameters
        written without a token (an optimizer step, load_state_dict) are still caught by the version check in program()."""
        sets = self.__dict__.setdefault("_weight_sets", {})
        cur = getattr(self, "_weight_token", None)
        if cur is not None and self._packed is not None and cur != token:
            sets[cur] = (self._packed, self._sd, self._programs, self._emb_off, self._emb_total, self._ln_flag, self.ln_unfolded)
        hit = sets.pop(token, None)
        if hit is not None and hit[6] == self.ln_unfolded:
            self._packed, self._sd, self._programs, self._emb_off, self._emb_total, self._ln_flag, _ = hit
            self._ctx_sig = None
        else:
            self.pack_weights()
        self._pack_sig = self._signature()
        self._weight_token = token
        while len(sets) > 1:                      # keep the set in use + one other
            sets.pop(next(iter(sets)))

    def _walk(self):
        for i, blk in enumerate(self.input_blocks):
            for j, m in enumerate(blk.layers):
                yield f"input_blocks.{i}.{j}.", m
        for j, m in enumerate(self.middle_block.layers):
            yield f"middle_block.{j}.", m
        for i, blk in enumerate(self.output_blocks):
            for j, m in enumerate(blk.layers):
                yield f"output_blocks.{i}.{j}.", m

    # ---- program construction ---------------------------------------------------------------------
    def _build(self, n, H, W_, L_ctx, c_concat, policy_n):
        P, sd = self._packed, self._sd
        dev = next(self.parameters()).device
        if self.resblock_updown and (H % (1 << (len(self.channel_mult) - 1)) or W_ % (1 << (len(self.channel_mult) - 1))):
            raise L.LdmkError(f"UNetModel(resblock_updown): {H}x{W_} input -- avg_pool2d(2, 2) of an odd grid drops its last row / column "
                              "in the reference and the skip shapes stop matching on the way up; use even sizes at every level")
        pg = Program(dev)
        pg.h2_flag = None          # (set per site by NetBuilder.site: engine.Program.plan runs a shape in F16X2 while it is a flag word)
        pg.far_plans = far_from_tuned(policy_n)      # a job far from every tuned batch carries the nearest tuned plans (engine.choose)
        mc = self.model_channels
        emb_ch = 4 * mc
        cx = self.in_channels - c_concat
        x_in = pg.alloc(n, cx, H, W_)
        cc_in = pg.alloc(n, c_concat, H, W_) if c_concat else None
        t_in = pg.alloc(n, dtype=torch.int64)
        ctx_in = pg.alloc(n * L_ctx, self.context_dim) if L_ctx else None      # (unconditional UNet: no context)
        y_emb = pg.alloc(n, emb_ch) if self.num_classes is not None else None     # label_emb(y) rows, gathered by forward()
        pg.inputs = dict(x=x_in, c_concat=cc_in, t=t_in, context=ctx_in, y_emb=y_emb)
        ctx_pg = Program(dev)     # context-only work: re-run only when the context changes
        ctx_pg._all = pg._all     # share accounting
        p_ = lambda t: 0 if t is None else t.data_ptr()
        pin = (policy_n, n)

        # -- time embedding MLP + all emb_layers (one launch each)
        temb = pg.alloc(n, mc)
        pg.add("ldmk_timestep_embedding", p_(t_in), p_(P["freqs"]), p_(temb), n, mc)
        e1 = pg.alloc(n, emb_ch)
        pg.add("ldmk_dense_small", p_(temb), mc, p_(P["te0"]), p_(sd["time_embed.0.bias"]), p_(e1), emb_ch, n, mc, emb_ch, 0)
        emb = pg.alloc(n, emb_ch)
        pg.add("ldmk_dense_small", p_(e1), emb_ch, p_(P["te2"]), p_(sd["time_embed.2.bias"]), p_(emb), emb_ch, n, emb_ch, emb_ch, 1)
        if y_emb is not None:
            pg.add("ldmk_axpy", p_(emb), p_(y_emb), 1.0, n * emb_ch)              # emb = emb + label_emb(y), openaimodel.py:726-728
        emb_all = pg.alloc(n, self._emb_total)
        pg.add("ldmk_dense_small", p_(emb), emb_ch, p_(P["emb_all"]), p_(P["emb_all_b"]), p_(emb_all), self._emb_total, n,
               emb_ch, self._emb_total, 1)

        # F16X2 where a kernel offers it, decided per SITE (engine.ArithSites): a site whose range flag went up runs in bf16x3
        nb_ = NetBuilder(pg, n, pin, sites=self._sites if self.f16x2 else None)
        psfx = lambda: "#p2" if nb_.h2_flag is not None else "#p"      # pre-split weight copies in the form of the site's arithmetic
        gn, conv, lin = nb_.gn, nb_.conv, nb_.lin

        def res_block(prefix, m, x0, x1, h, w):
            hw = h * w
            # GroupNorm+SiLU materialised once (stats pass + one elementwise pass over the concat); the conv
            # then reads it raw -- cheaper than re-normalising every element 9 x (N/tile) times in the gather
            # (gn_conv: one elementwise GroupNorm+SiLU pass + implicit-GEMM conv, or the Winograd route for the wide levels)
            bv = emb_all.data_ptr() + 4 * self._emb_off[prefix]
            # use_scale_shift_norm (openaimodel.py:267-271): the embedding does not add to conv1's output but modulates the second
            # GroupNorm -- h = norm(h) (1 + scale) + shift -- which is a per-sample edit of that norm's coefficient planes
            film = (bv, self._emb_total) if m.scale_shift else None
            if m.updown:
                return res_block_updown(prefix, m, x0, h, w, bv, film)
            with nb_.site(prefix + "in_layers"):
                h1 = nb_.gn_conv(x0, x1, h, w, sd[prefix + "in_layers.0.weight"], sd[prefix + "in_layers.0.bias"], 1e-5,
                                 P[prefix + "c1"], P.get(prefix + "c1#wg"), sd[prefix + "in_layers.2.bias"],
                                 batch_vec=None if film else bv,
                                 bv_ld=self._emb_total, stats=True, wf=P.get(prefix + "c1#f"), u_ps=P.get(prefix + "c1#wg" + psfx()),
                                 wp_ps=P.get(prefix + "c1#pc2"))
            g2, b2 = sd[prefix + "out_layers.0.weight"], sd[prefix + "out_layers.0.bias"]
            if m.cin != m.cout:
                x0r = x0.reshape(n * hw, -1)
                x1r = None if x1 is None else x1.reshape(n * hw, -1)
                with nb_.site(prefix + "skip_connection"):
                    skip = lin(x0r, P[prefix + "skip"], sd[prefix + "skip_connection.bias"], hw, x1=x1r, wf=P.get(prefix + "skip#f"))
                with nb_.site(prefix + "out_layers"):
                    out = nb_.gn_conv(h1, None, h, w, g2, b2, 1e-5, P[prefix + "c2"], P.get(prefix + "c2#wg"),
                                      sd[prefix + "out_layers.3.bias"], residual=skip, out=skip.view(n, h, w, m.cout), stats=True,
                                      wf=P.get(prefix + "c2#f"), u_ps=P.get(prefix + "c2#wg" + psfx()), wp_ps=P.get(prefix + "c2#pc2"), film=film)
            else:
                assert x1 is None
                with nb_.site(prefix + "out_layers"):
                    out = nb_.gn_conv(h1, None, h, w, g2, b2, 1e-5, P[prefix + "c2"], P.get(prefix + "c2#wg"),
                                      sd[prefix + "out_layers.3.bias"], residual=x0, stats=True, wf=P.get(prefix + "c2#f"),
                                      u_ps=P.get(prefix + "c2#wg" + psfx()), wp_ps=P.get(prefix + "c2#pc2"), film=film)
            nb_.release(h1)
            return out

        def res_block_updown(prefix, m, x, h, w, bv, film):
            """ResBlock(up=True / down=True) (resblock_updown, openaimodel.py:256-261): SiLU(GroupNorm(x)) and x itself go through
            the parameter-free Upsample (nearest x2) / Downsample (avg_pool2d(2, 2)) -- one ldmk_resample2 pass each -- before the
            first convolution and on the skip path; cin == cout, so the skip is the identity.  The first convolution reads the
            resampled activation raw (direct implicit GEMM), the second half is the plain block's."""
            up = m.updown == "up"
            assert m.cin == m.cout and (up or (h % 2 == 0 and w % 2 == 0))      # (out_channels == channels, openaimodel.py:573,663)
            oh, ow = (2 * h, 2 * w) if up else (h // 2, w // 2)
            sh_, sw_ = (h, w) if up else (oh, ow)              # ldmk_resample2 takes the SMALLER grid
            a = nb_.gn_act(x, None, h * w, sd[prefix + "in_layers.0.weight"], sd[prefix + "in_layers.0.bias"], 1e-5)
            a_r = pg.alloc(n, oh, ow, m.cin)
            pg.add("ldmk_resample2", p_(a), p_(a_r), n, sh_, sw_, m.cin, 1 if up else 0)
            nb_.release(a)
            x_r = pg.alloc(n, oh, ow, m.cin)
            pg.add("ldmk_resample2", p_(x), p_(x_r), n, sh_, sw_, m.cin, 1 if up else 0)
            with nb_.site(prefix + "in_layers"):
                h1 = conv(a_r, None, P[prefix + "c1"], sd[prefix + "in_layers.2.bias"], oh, ow, batch_vec=None if film else bv,
                          bv_ld=self._emb_total, stats=True, wf=P.get(prefix + "c1#f"))
            nb_.release(a_r)
            with nb_.site(prefix + "out_layers"):
                out = nb_.gn_conv(h1, None, oh, ow, sd[prefix + "out_layers.0.weight"], sd[prefix + "out_layers.0.bias"], 1e-5,
                                  P[prefix + "c2"], P.get(prefix + "c2#wg"), sd[prefix + "out_layers.3.bias"], residual=x_r,
                                  out=x_r, stats=True, wf=P.get(prefix + "c2#f"), u_ps=P.get(prefix + "c2#wg" + psfx()),
                                  wp_ps=P.get(prefix + "c2#pc2"), film=film)
            nb_.release(h1)
            return out

        def spatial_tf(prefix, m, x, h, w):
            return emit_spatial_transformer(nb_, ctx_pg, P, sd, prefix, m, x, h, w, L_ctx, ctx_in, self.context_dim,
                                            unfolded=self.ln_unfolded, ln_flag=self._ln_flag)

        def run_layers(prefix, layers, x0, x1, h, w, keep_input):
            """Returns (out, h, w). Releases intermediate tensors that no skip connection holds."""
            cur0, cur1 = x0, x1
            owned = not keep_input
            for j, m in enumerate(layers):
                p = f"{prefix}{j}."
                if m.kind == "res":
                    out = res_block(p, m, cur0, cur1, h, w)
                    if m.updown:
                        h, w = (2 * h, 2 * w) if m.updown == "up" else (h // 2, w // 2)
                elif m.kind == "st":
                    out = spatial_tf(p, m, cur0, h, w)
                elif m.kind == "attn":
                    out = emit_attention_block(nb_, P, sd, p, m, cur0, h, w)
                elif m.kind == "down":
                    with nb_.site(p + "op"):
                        out = conv(cur0, None, P[p + "w"], sd[p + "op.bias"], h, w, stride=2, stats=True)
                    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                elif m.kind == "up":
                    with nb_.site(p + "conv"):
                        out = nb_.up_conv(cur0, h, w, P[p + "w"], P.get(p + "w#up"), sd[p + "conv.bias"], stats=True, w4_ps=P.get(p + "w#up" + psfx()))
                    h, w = 2 * h, 2 * w
                else:
                    raise AssertionError(m.kind)
                if owned:
                    nb_.release(cur0)      # cur1 (a skip tensor) is released by the caller
                cur0, cur1, owned = out, None, True
            return cur0, h, w

        # ---- the UNet walk (openaimodel.py:729-742)
        hs = []
        h0 = pg.alloc(n, H, W_, mc)
        pg.add("ldmk_conv3x3_in", p_(x_in), cx, p_(cc_in), c_concat, p_(P["input_blocks.0.0.w"]),
               p_(sd["input_blocks.0.0.bias"]), p_(h0), n, H, W_, mc)
        hcur, ch_, cw_ = h0, H, W_
        hs.append((hcur, ch_, cw_))
        for i in range(1, len(self.input_blocks)):
            hcur, ch_, cw_ = run_layers(f"input_blocks.{i}.", self.input_blocks[i].layers, hcur, None, ch_, cw_, True)
            hs.append((hcur, ch_, cw_))
        hcur, ch_, cw_ = run_layers("middle_block.", self.middle_block.layers, hcur, None, ch_, cw_, True)
        for i, blk in enumerate(self.output_blocks):
            skip, sh, sw = hs.pop()
            assert (sh, sw) == (ch_, cw_)
            new, ch_, cw_ = run_layers(f"output_blocks.{i}.", blk.layers, hcur, skip, ch_, cw_, True)
            nb_.release(hcur, skip)     # both inputs of the concat are dead once the block has run
            hcur = new
        coef = gn(hcur, None, ch_ * cw_, sd["out.0.weight"], sd["out.0.bias"], 1e-5)
        eps = pg.alloc(n, self.out_channels, H, W_)
        # 4x4-pixel workgroups with the weights in LDS while the 16x16-pixel form would leave CUs without a workgroup (16x the
        # workgroups: 60 -> ~15 us at 32x32, B = 16); from one 16x16-pixel workgroup per CU up the larger tile wins (64x64, B = 16:
        # 58 us against 166 us, r02 / r03 layer tables).  Decided on the plan-policy batch like every plan, so a sample's result
        # does not depend on how its job is sharded.
        co_small = (policy_n * H * W_ < 256 * 256 and self._final_ch % 4 == 0
                    and 9 * self._final_ch * self.out_channels * 4 <= 60 * 1024)
        pg.add("ldmk_conv3x3_out_small" if co_small else "ldmk_conv3x3_out", p_(hcur), p_(coef), p_(P["out"]), p_(sd["out.2.bias"]),
               p_(eps), n, H, W_, self._final_ch, self.out_channels)
        pg.outputs = dict(eps=eps)
        pg.ctx_program = ctx_pg
        return pg

    # ---- public surface ---------------------------------------------------------------------------
    # how many evaluations in a row may answer a raised range flag by denying only the raised SITES before the whole model goes
    # back to bf16x3.  Out-of-range operands are saturated, not overflowed (csrc/ldmk_common.h), so one pass normally names
    # every site that has to change; a second round catches sites that only show their range once their inputs are right.
    H2_SITE_PASSES = 2

    def flags_tripped(self, check_layernorm=True):
        """Host read (one sync) of the device flags the launch programs raise; True if whatever was just computed has to be computed
        again.  (1) F16X2 range flags, one word per site (engine.ArithSites): the raised sites are denied that arithmetic -- they run
        in bf16x3 from the next program on, everything else stays in F16X2 -- and after H2_SITE_PASSES such rounds in a row the
        whole model goes back to bf16x3.  (2) The folded-LayerNorm statistics passes: rows with |mean| > LN_GUARD_RATIO standard
        deviations -- the model switches to the unfolded prologue for good (weights re-packed).  Programs are dropped either way."""
        import warnings
        if self.f16x2 and self._sites is not None:
            raised = self._sites.raised()
            if raised:
                self._h2_passes += 1
                if self._h2_passes > self.H2_SITE_PASSES:
                    warnings.warn("UNetModel: operands keep leaving the range of the F16X2 arithmetic (|x| >= 1000) after "
                                  f"{self.H2_SITE_PASSES} rounds of per-site fall-back; switching the whole model to the bf16x3 "
                                  "arithmetic (LDMK_F16X2=0 starts there)", RuntimeWarning, stacklevel=3)
                    self.f16x2 = False
                else:
                    self._sites.denied.update(raised)
                    warnings.warn(f"UNetModel: an operand left the range of the F16X2 arithmetic (|x| >= 1000) at {len(raised)} of "
                                  f"{len(self._sites.index)} sites ({', '.join(raised[:4])}{', ...' if len(raised) > 4 else ''}); "
                                  "these run in the bf16x3 arithmetic from now on, the others stay in F16X2", RuntimeWarning, stacklevel=3)
                if self._ln_flag is not None:
                    self._ln_flag.zero_()       # (statistics of that run may be those of saturated operands: the repeat decides afresh)
                self._programs.clear()
                for st_ in self.__dict__.get("_weight_sets", {}).values():
                    st_[2].clear()              # (the stored weight sets' programs were planned with the old denials too)
                self.generation += 1
                return True
            self._h2_passes = 0
        if check_layernorm and not (self.ln_unfolded or self._ln_flag is None or int(self._ln_flag.item()) == 0):
            warnings.warn(f"UNetModel: token rows with |mean| > {LN_GUARD_RATIO:g} standard deviations reached a LayerNorm; the folded "
                          "form (LayerNorm through the product) loses accuracy on them -- switching this model to the unfolded "
                          "prologue (LDMK_LN_UNFOLDED=1 starts there)", RuntimeWarning, stacklevel=3)
            self.ln_unfolded = True
            tok = getattr(self, "_weight_token", None)
            self.__dict__.get("_weight_sets", {}).clear()
            self.pack_weights()
            self._weight_token = tok
            return True
        return False

    layernorm_guard_tripped = flags_tripped       # (the name rounds 2-4 used)

    def deny_f16x2(self, names):
        """Run the named sites (arithmetic_status()['denied'] of an earlier session, say) in bf16x3 from the start, so that a
        checkpoint known to carry large activations there does not pay for finding out again."""
        names = set(names)
        self._denied0 |= names
        if self._sites is not None and not names <= self._sites.denied:
            self._sites.denied |= names
            self._programs.clear()
            for st_ in self.__dict__.get("_weight_sets", {}).values():
                st_[2].clear()
            self.generation += 1

    def arithmetic_status(self):
        """{'f16x2': model-wide switch, 'sites': F16X2 sites seen so far, 'denied': names of the sites that run in bf16x3 after a
        raised range flag, 'flags_up': names whose flag is up right now (one host sync, nothing is cleared), 'ln_unfolded'}."""
        up = [] if self._sites is None else self._sites.raised(clear=False)
        return {"f16x2": bool(self.f16x2), "sites": 0 if self._sites is None else len(self._sites.index),
                "denied": sorted(self._sites.denied) if self._sites is not None else [], "flags_up": up,
                "ln_flag_up": bool(self._ln_flag is not None and int(self._ln_flag.item()) != 0), "ln_unfolded": bool(self.ln_unfolded)}

    def program(self, n, H, W_, L_ctx, c_concat=0):
        if self._packed is None or (self.auto_repack and self._pack_sig != self._signature()):
            self.pack_weights()
        policy_n = self.policy_batch or n
        key = (n, H, W_, L_ctx, c_concat, policy_n)
        pg = self._programs.get(key)
        if pg is None:
            from . import unet_small
            plain = self._heads32 and not self.use_scale_shift_norm and self.num_classes is None and not self.resblock_updown
            if plain and unet_small.wants_small_route(policy_n, H, W_, L_ctx):
                # batch 1-2 (the reference's shipped talking-face mode): the program cut for few dependent launches
                pg = unet_small.build_small(self, n, H, W_, L_ctx, c_concat, policy_n)
            else:
                pg = self._build(n, H, W_, L_ctx, c_concat, policy_n)
            self._programs[key] = pg
        return pg

    def convert_to_fp16(self):
        """openaimodel.py:694-700 casts the torso to float16; this path is fp32 storage with fp32-class products."""
        raise NotImplementedError("UNetModel.convert_to_fp16: fp32 only (reference precision; use_fp16 is refused the same way)")

    def convert_to_fp32(self):
        """openaimodel.py:702-708: the torso back to float32 -- it never leaves it here."""
        return None

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, c_concat=None, **kwargs):
        """x: (N, C, H, W) fp32 NCHW; timesteps: (N,) int64; context: (N, L, context_dim).
        `c_concat` (N, C2, H, W) is concatenated to x on the channel axis inside the first conv (the TF
        DiffusionWrapper does `torch.cat([x] + c_concat, 1)`, ddpm2cond.py:1309); passing an already
        concatenated x works too."""
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"   # openaimodel.py:720-722
        if context is None and self.use_spatial_transformer:
            raise L.LdmkError("UNetModel.forward: context is required (the reference raises a shape error for "
                              "context=None when context_dim != inner dim, attention.py:174-175)")
        if context is not None and not self.use_spatial_transformer:
            raise L.LdmkError("UNetModel.forward: this UNet has no cross-attention (use_spatial_transformer=False): context must be None")
        if not x.is_cuda:
            raise L.LdmkError("UNetModel.forward: input must be a CUDA tensor (no CPU fallback)")
        n, cx, H, W_ = x.shape
        cc = 0 if c_concat is None else c_concat.shape[1]
        assert cx + cc == self.in_channels, f"got {cx}+{cc} input channels, model has {self.in_channels}"
        assert context is None or (context.shape[0] == n and context.shape[2] == self.context_dim)
        L_ctx = 0 if context is None else context.shape[1]
        for _ in range(MAX_ARITHMETIC_PASSES):
            pg = self.program(n, H, W_, L_ctx, cc)
            pg.inputs["x"].copy_(x)
            if cc:
                pg.inputs["c_concat"].copy_(c_concat)
            pg.inputs["t"].copy_(timesteps.to(torch.int64))
            if L_ctx:
                pg.inputs["context"].copy_(context.reshape(n * L_ctx, self.context_dim))
            if y is not None:
                assert y.shape == (n,)
                pg.inputs["y_emb"].copy_(self.label_emb.weight.detach().float()[y.to(torch.int64)])
            pg.ctx_program.run()
            pg.run()
            if torch.cuda.is_current_stream_capturing():
                break
            # The F16X2 range flags are DATA dependent (an activation of 1000 can show up in any call), so EVERY evaluation reads
            # them -- one small device -> host copy next to the clone below; a raised site is re-planned in bf16x3 and the
            # evaluation repeated.  The folded-LayerNorm guard (an accuracy matter, not an overflow) is read on the first
            # evaluation of every program only; the samplers read both once per run instead.
            first = not getattr(pg, "ln_checked", False)
            pg.ln_checked = True
            if not self.flags_tripped(check_layernorm=first):
                break
        return pg.outputs["eps"].clone()
