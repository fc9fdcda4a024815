"""Tensor-level wrappers of the training-step kernels (include/ldmk.h, section "Training step").
CUDA tensors only; every call goes to libldmk.so -- there is no PyTorch fallback."""
import ctypes as C

import torch

from . import lib as L
from . import ops
from .ops import _ptr, stream


def _f32(*shape, device="cuda"):
    return torch.empty(*shape, device=device, dtype=torch.float32)


# Arithmetic of the matrix-core GEMMs the training helpers below launch: L.COMPUTE_F32 (parity path) or L.COMPUTE_BF16
# (BASELINE configs[4]: bf16 operands, fp32 accumulation / master weights).  UNetTrainer sets it from its `compute` argument
# at the start of every forward / backward; the step is single-threaded host code.
COMPUTE = L.COMPUTE_F32


def set_compute(mode):
    global COMPUTE
    COMPUTE = {None: L.COMPUTE_F32, "f32": L.COMPUTE_F32, "fp32": L.COMPUTE_F32, "bf16": L.COMPUTE_BF16}.get(mode, mode)
    return COMPUTE


def wgrad_args(R, Kw, N, a, dy, dw, c=0, lda=None, conv=None, ldy=None, ldw=None, accumulate=False, alpha=1.0, batch=1,
               a_bstride=0, dy_bstride=0, dw_bstride=0, splitr=0, ws=None, dbias=None):
    w = L.WgradArgs()
    w.R, w.Kw, w.N = R, Kw, N
    w.a, w.dy, w.dw = _ptr(a), _ptr(dy), _ptr(dw)
    w.c = c or Kw
    if conv is not None:
        w.a_mode = L.A_CONV3X3
        w.in_h, w.in_w, w.out_h, w.out_w, w.stride, w.pad_lo, w.upsample = conv
        w.lda = w.c
    else:
        w.a_mode = L.A_ROWS
        w.lda = lda if lda is not None else Kw
        w.out_h, w.out_w = 1, 1
    w.ldy = ldy if ldy is not None else N
    w.ldw = ldw if ldw is not None else N
    w.accumulate, w.alpha = 1 if accumulate else 0, alpha
    w.batch, w.a_bstride, w.dy_bstride, w.dw_bstride = batch, a_bstride, dy_bstride, dw_bstride
    w.splitr = splitr
    w.dbias = _ptr(dbias)
    w.compute = COMPUTE
    if ws is not None:
        w.ws, w.ws_elems = ws.data_ptr(), ws.numel()
    return w


def wgrad_workspace_elems(w):
    sr = C.c_int(0)
    L.call("ldmk_wgrad_plan", C.byref(w), C.byref(sr))
    return sr.value, (sr.value * max(1, w.batch) * (w.Kw + (1 if w.dbias else 0)) * w.N if sr.value > 1 else 0)


def wgrad(w):
    L.call("ldmk_wgrad", C.byref(w), stream())


_WS = {}
_WS_RETIRED = []     # outgrown scratches stay alive: a captured hipGraph may still replay partial sums into them


def _workspace(dev, elems):
    """One scratch per device for the wgrad partial slabs (stream-ordered reuse).  It only grows outside a stream
    capture, geometrically (so the retired ones add up to less than the live one), and an outgrown scratch is never
    freed: hipGraphs captured earlier keep its address."""
    ws = _WS.get(dev)
    if ws is None or ws.numel() < elems:
        if torch.cuda.is_current_stream_capturing():
            raise L.LdmkError(f"wgrad scratch of {elems} floats requested during a stream capture, but only "
                              f"{0 if ws is None else ws.numel()} are allocated: run the step once eagerly first")
        if ws is not None:
            _WS_RETIRED.append(ws)
        ws = _f32(max(int(elems), 1 << 22, 0 if ws is None else 2 * ws.numel()), device=dev)
        _WS[dev] = ws
    return ws


def wgrad_linear(a2d, dy2d, dw=None, accumulate=False, lda=None, splitr=0, dbias=None):
    """dW[K][N] = a2d[R][K]^T @ dy2d[R][N]  (+ dbias[N] = column sums of dy2d)."""
    R, K = a2d.shape
    N = dy2d.shape[1]
    if dw is None:
        dw = _f32(K, N, device=a2d.device)
    w = wgrad_args(R, K, N, a2d, dy2d, dw, lda=lda if lda is not None else a2d.stride(0), ldy=dy2d.stride(0),
                   accumulate=accumulate, splitr=splitr, dbias=dbias)
    sr, need = wgrad_workspace_elems(w) if splitr == 0 else (splitr, splitr * (K + 1) * N)
    ws = _workspace(a2d.device, need)
    w.splitr, w.ws, w.ws_elems = sr, ws.data_ptr(), ws.numel()
    wgrad(w)
    return dw


def wgrad_conv3x3(x, dy, stride=1, pad_lo=1, upsample=False, dw=None, accumulate=False, dbias=None):
    """x: (n,h,w,c) NHWC input the forward conv consumed, dy: (n,oh,ow,cout) -> packed dW [9c][cout]."""
    n, h, w_, c = x.shape
    _, oh, ow, cout = dy.shape
    if dw is None:
        dw = _f32(9 * c, cout, device=x.device)
    w = wgrad_args(n * oh * ow, 9 * c, cout, x, dy, dw, c=c, conv=(h, w_, oh, ow, stride, pad_lo, 1 if upsample else 0),
                   accumulate=accumulate, dbias=dbias)
    sr, need = wgrad_workspace_elems(w)
    ws = _workspace(x.device, need)
    w.splitr, w.ws, w.ws_elems = sr, ws.data_ptr(), ws.numel()
    wgrad(w)
    return dw


def pack_dgrad3x3(wp, cin, cout, out=None):
    if out is None:
        out = _f32(9 * cout, cin, device=wp.device)
    L.call("ldmk_pack_dgrad3x3", _ptr(wp), _ptr(out), cin, cout, stream())
    return out


def conv3x3_dgrad(dy, wd, in_hw, stride=1, out=None, residual=None):
    """Data gradient of a pad-1 3x3 convolution: dy (n,oh,ow,cout), wd = pack_dgrad3x3 weights [9*cout][cin] ->
    (n,h,w,cin).  stride 2 reads dy zero-inserted (ldmk_igemm upsample=2)."""
    n, oh, ow, cout = dy.shape
    h, w_ = in_hw
    cin = wd.shape[1]
    if out is None:
        out = _f32(n, h, w_, cin, device=dy.device)
    a = ops.make_igemm_args(n * h * w_, cin, 9 * cout, dy, cout, wd, out, cin, h * w_, compute=COMPUTE,
                            conv=(oh, ow, h, w_, 1, 1, 2 if stride == 2 else 0), residual=residual)
    ops.igemm(a)
    return out


def gn_group_stats(partial0, c0, partial1, c1, n, hw, groups, eps, out=None):
    if out is None:
        out = _f32(n, groups, 2, device=partial0.device)
    L.call("ldmk_gn_group_stats", _ptr(partial0), c0, _ptr(partial1), c1, n, hw, groups, eps, _ptr(out), stream())
    return out


def gn_bwd(x0, x1, dy, coef, mr, gamma, n, hw, groups=32, silu=True, dx0=None, acc0=False, dx1=None, acc1=False,
           dgamma=None, dbeta=None, acc_params=False):
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    Cc = c0 + c1
    dev = x0.device
    dx0 = torch.empty_like(x0) if dx0 is None else dx0
    if x1 is not None and dx1 is None:
        dx1 = torch.empty_like(x1)
    dgamma = _f32(Cc, device=dev) if dgamma is None else dgamma
    dbeta = _f32(Cc, device=dev) if dbeta is None else dbeta
    scratch = _f32(L.load().ldmk_gn_bwd_scratch_elems(n, hw, Cc, groups), device=dev)
    L.call("ldmk_gn_bwd", _ptr(x0), c0, _ptr(x1), c1, _ptr(dy), _ptr(coef), _ptr(mr), _ptr(gamma), n, hw, groups,
           1 if silu else 0, _ptr(dx0), 1 if acc0 else 0, _ptr(dx1), 1 if acc1 else 0, _ptr(dgamma), _ptr(dbeta),
           1 if acc_params else 0, _ptr(scratch), stream())
    return dx0, dx1, dgamma, dbeta


def ln_apply(x2d, stats, gamma, beta, out=None):
    rows, c = x2d.shape
    out = torch.empty_like(x2d) if out is None else out
    L.call("ldmk_ln_apply", _ptr(x2d), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(out), rows, c, stream())
    return out


def ln_bwd(dy, x2d, stats, gamma, dx=None, acc_dx=False, dgamma=None, dbeta=None, acc_params=False):
    rows, c = x2d.shape
    dev = x2d.device
    dx = torch.empty_like(x2d) if dx is None else dx
    dgamma = _f32(c, device=dev) if dgamma is None else dgamma
    dbeta = _f32(c, device=dev) if dbeta is None else dbeta
    scratch = _f32(L.load().ldmk_ln_bwd_blocks(rows) * c * 2, device=dev)
    L.call("ldmk_ln_bwd", _ptr(dy), _ptr(x2d), _ptr(stats), _ptr(gamma), _ptr(dx), 1 if acc_dx else 0, rows, c, _ptr(dgamma),
           _ptr(dbeta), 1 if acc_params else 0, _ptr(scratch), stream())
    return dx, dgamma, dbeta


def geglu_fwd(pre, out=None):
    rows, two = pre.shape
    out = _f32(rows, two // 2, device=pre.device) if out is None else out
    L.call("ldmk_geglu_fwd", _ptr(pre), _ptr(out), rows, two // 2, stream())
    return out


def geglu_bwd(pre, df, out=None):
    rows, two = pre.shape
    out = torch.empty_like(pre) if out is None else out
    L.call("ldmk_geglu_bwd", _ptr(pre), _ptr(df), _ptr(out), rows, two // 2, stream())
    return out


def softmax_bwd_rows_(p2d, dp2d, scale=1.0):
    rows, cols = p2d.shape
    L.call("ldmk_softmax_bwd_rows", _ptr(p2d), _ptr(dp2d), rows, cols, scale, stream())
    return dp2d


def colsum(x2d, rows_per_group=None, out=None, accumulate=False):
    rows, n = x2d.shape
    rpg = rows if rows_per_group is None else rows_per_group
    groups = rows // rpg
    out = _f32(groups, n, device=x2d.device) if out is None else out
    scratch = _f32(groups * L.load().ldmk_colsum_splits(rpg) * n, device=x2d.device)
    L.call("ldmk_colsum", _ptr(x2d), x2d.stride(0), rpg, groups, n, _ptr(out), out.stride(0) if out.dim() > 1 else n,
           1 if accumulate else 0, _ptr(scratch), stream())
    return out


def sumpool2(du, out=None, accumulate=False):
    n, h2, w2, c = du.shape
    out = _f32(n, h2 // 2, w2 // 2, c, device=du.device) if out is None else out
    L.call("ldmk_sumpool2", _ptr(du), _ptr(out), n, h2 // 2, w2 // 2, c, 1 if accumulate else 0, stream())
    return out


def silu(x, out=None):
    out = torch.empty_like(x) if out is None else out
    L.call("ldmk_silu", _ptr(x), _ptr(out), x.numel(), stream())
    return out


def silu_bwd(x, dy, out=None):
    out = torch.empty_like(x) if out is None else out
    L.call("ldmk_silu_bwd", _ptr(x), _ptr(dy), _ptr(out), x.numel(), stream())
    return out


def axpy_(y, x, a=1.0):
    L.call("ldmk_axpy", _ptr(y), _ptr(x), a, y.numel(), stream())
    return y


def q_sample(x0, noise, t, sqrt_ac, sqrt_1mac, out=None):
    out = torch.empty_like(x0) if out is None else out
    L.call("ldmk_q_sample", _ptr(x0), _ptr(noise), _ptr(t), _ptr(sqrt_ac), _ptr(sqrt_1mac), _ptr(out), x0.shape[0],
           x0[0].numel(), stream())
    return out


def mse_grad(pred, target, dpred=None, loss=None, denom=0):
    dpred = torch.empty_like(pred) if dpred is None else dpred
    loss = _f32(1, device=pred.device) if loss is None else loss
    scratch = torch.empty(256, device=pred.device, dtype=torch.float64)
    L.call("ldmk_mse_grad", _ptr(pred), _ptr(target), _ptr(dpred), pred.numel(), denom, _ptr(loss), _ptr(scratch), stream())
    return loss, dpred


def adamw_(p, g, m, v, lr, betas, eps, weight_decay, step):
    L.call("ldmk_adamw", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step, stream())


def ema_(shadow, p, one_minus_decay):
    L.call("ldmk_ema", _ptr(shadow), _ptr(p), p.numel(), one_minus_decay, stream())


def head_permute(src, n, tokens, parts, heads, to_heads, out=None):
    out = torch.empty_like(src) if out is None else out
    L.call("ldmk_head_permute", _ptr(src), _ptr(out), n, tokens, parts, heads, 1 if to_heads else 0, stream())
    return out


def attention_backward(qkv, datt, n, tokens, heads):
    """Gradient of ldmk_attn_self w.r.t. the fused qkv rows: scores re-materialised per head ([n*heads][T][T]) and the
    five products run as batched GEMMs on the matrix cores (attention.py:178-192 backward)."""
    Z, T_, scale = n * heads, tokens, 32 ** -0.5
    dev = qkv.device
    h = head_permute(qkv, n, T_, 3, heads, True).view(3, Z, T_, 32)
    q, k, v = h[0], h[1], h[2]
    do = head_permute(datt, n, T_, 1, heads, True).view(Z, T_, 32)
    p = ops.bmm(q, k, True, alpha=scale)                               # [Z][T][T]
    ops.softmax_rows_(p.view(Z * T_, T_), 1.0)
    dh = torch.empty(3, Z, T_, 32, device=dev)
    _wgrad_batched(p, do, dh[2], Z, T_, T_, 32)                         # dV = P^T dO
    dp = ops.bmm(do, v, True)                                          # dP = dO V^T
    softmax_bwd_rows_(p.view(Z * T_, T_), dp.view(Z * T_, T_), scale)   # dS (scaled)
    ops.bmm(dp, k, False, out=dh[0])                                   # dQ = dS K
    _wgrad_batched(dp, q, dh[1], Z, T_, T_, 32)                         # dK = dS^T Q
    return head_permute(dh, n, T_, 3, heads, False).view(n * T_, 3 * heads * 32)


def _wgrad_batched(a, dy, out, Z, R, Kw, N):
    w = wgrad_args(R, Kw, N, a, dy, out, batch=Z, a_bstride=R * Kw, dy_bstride=R * N, dw_bstride=Kw * N)
    sr, need = wgrad_workspace_elems(w)
    ws = _workspace(a.device, need)
    w.splitr, w.ws, w.ws_elems = sr, ws.data_ptr(), ws.numel()
    wgrad(w)


def attn_self_lse(qkv, n, tokens, heads):
    """Forward attention that also returns the per-row log-sum-exp [n][heads][tokens] (saved for the backward)."""
    out = _f32(n * tokens, heads * 32, device=qkv.device)
    lse = _f32(n, heads, tokens, device=qkv.device)
    # bf16 compute mode: Q K^T and P V on the bf16 matrix cores (fp32 softmax / statistics / storage), like the GEMMs
    name = "ldmk_attn_self_lse_bf16" if COMPUTE == L.COMPUTE_BF16 else "ldmk_attn_self_lse"
    L.call(name, _ptr(qkv), _ptr(out), _ptr(lse), n, tokens, heads, 32 ** -0.5, stream())
    return out, lse


def attn_self_bwd(qkv, out, dout, lse, n, tokens, heads):
    """d(qkv) of ldmk_attn_self, flash style (probabilities recomputed from lse; any token count)."""
    dqkv = torch.empty_like(qkv)
    dsum = _f32(n * heads * tokens, device=qkv.device)
    name = "ldmk_attn_self_bwd_bf16" if COMPUTE == L.COMPUTE_BF16 else "ldmk_attn_self_bwd"
    L.call(name, _ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), _ptr(dsum), n, tokens, heads, 32 ** -0.5, stream())
    return dqkv


def attn_cross_bwd(q, k, v, dout, n, tokens, ctx_len, heads):
    """Gradients of ldmk_attn_cross: q [n*tokens][C], k / v [n*ctx_len][C], dout [n*tokens][C] -> (dq, dk, dv)."""
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    scratch = _f32(2 * n * tokens * heads * ctx_len, device=q.device)
    L.call("ldmk_attn_cross_bwd", _ptr(q), q.stride(0), _ptr(k), _ptr(v), k.stride(0), _ptr(dout), dout.stride(0), _ptr(dq),
           _ptr(dk), _ptr(dv), _ptr(scratch), n, tokens, ctx_len, heads, 32 ** -0.5, stream())
    return dq, dk, dv
