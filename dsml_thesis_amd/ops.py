"""Tensor-level wrappers over the C ABI (include/ldmk.h).  PyTorch is only the allocator and the
stream provider here; every function enqueues HIP kernels from libldmk.so on torch's current
stream and never synchronises.  Activations are NHWC float32 CUDA tensors."""
import ctypes as C
import math

import weakref

import torch

from . import lib as L


def _ptr(t):
    """device pointer of a tensor; None -> 0; an int is taken as a raw device pointer (e.g. an offset into a tensor)"""
    return 0 if t is None else (t if isinstance(t, int) else t.data_ptr())


def stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise L.LdmkError(f"{name}: expected a contiguous float32 CUDA tensor, got {t.dtype} {t.device} "
                          f"contiguous={t.is_contiguous()}")


# ------------------------------------------------------------------------------------------ packing
def pack_conv3x3(w):
    """torch OIHW (O,I,3,3) -> the igemm conv K order [I/32][9][32][O] (as a [9*I][O] matrix).  I % 32 == 0."""
    _chk(w, "pack_conv3x3")
    o, i, kh, kw = w.shape
    assert kh == 3 and kw == 3
    out = torch.empty(9 * i, o, device=w.device, dtype=torch.float32)
    L.call("ldmk_pack_conv3x3", _ptr(w), _ptr(out), o, i, stream())
    return out


def pack_conv3x3_narrow(w):
    """torch OIHW (O,I,3,3) -> [9][I][O] for the narrow boundary convs (ldmk_conv3x3_in / ldmk_conv3x3_out)."""
    _chk(w, "pack_conv3x3")
    o, i, kh, kw = w.shape
    assert kh == 3 and kw == 3
    out = torch.empty(9 * i, o, device=w.device, dtype=torch.float32)
    L.call("ldmk_permute3", _ptr(w), _ptr(out), o, i, 9, 2, 1, 0, stream())
    return out


def pack_linear(w):
    """torch [N][K] (or conv1x1 [N][K][1][1]) -> [K][N]."""
    _chk(w, "pack_linear")
    n, k = w.shape[0], w.shape[1]
    out = torch.empty(k, n, device=w.device, dtype=torch.float32)
    L.call("ldmk_permute3", _ptr(w), _ptr(out), n, k, 1, 1, 0, 2, stream())
    return out


def pack_wfrag(wp):
    """[K][N] packed weight -> its MFMA-fragment-order copy for the row GEMM (ldmk_pack_wfrag); None when the shape
    cannot be packed (K % 8 or N % 32)."""
    _chk(wp, "pack_wfrag")
    k, n = wp.shape
    if L.load().ldmk_wfrag_elems(k, n) < 0:
        return None
    out = torch.empty(k * n, device=wp.device, dtype=torch.float32)
    L.call("ldmk_pack_wfrag", _ptr(wp), n, k, n, _ptr(out), stream())
    return out


def pack_geglu(w, b):
    """GEGLU proj weight [2*inner][K] / bias [2*inner] -> packed [K][2*inner] / [2*inner] where every
    64-column block is (32 value columns | their 32 gate columns), so that one wave holds both halves."""
    two_inner, k = w.shape
    inner = two_inner // 2
    assert inner % 32 == 0
    wt = pack_linear(w)                                   # [K][2*inner]
    val = wt[:, :inner].reshape(k, inner // 32, 1, 32)
    gate = wt[:, inner:].reshape(k, inner // 32, 1, 32)
    wp = torch.cat([val, gate], dim=2).reshape(k, two_inner).contiguous()
    bp = torch.cat([b[:inner].reshape(-1, 1, 32), b[inner:].reshape(-1, 1, 32)], dim=1).reshape(-1).contiguous()
    return wp, bp


# ------------------------------------------------------------------------------------------ norms
def gn_coef(x0, x1, n, hw, gamma, beta, eps, groups=32, partial=None, coef=None):
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    C_ = c0 + c1
    chunks = L.load().ldmk_gn_chunks(hw)
    if partial is None:
        partial = torch.empty(n * chunks * C_ * 3, device=x0.device, dtype=torch.float32)
    if coef is None:
        coef = torch.empty(n, 2, C_, device=x0.device, dtype=torch.float32)
    L.call("ldmk_gn_coef", _ptr(x0), c0, _ptr(x1), c1, n, hw, groups, float(eps), _ptr(gamma), _ptr(beta),
           _ptr(partial), _ptr(coef), stream())
    return coef


def gn_apply(x0, x1, coef, n, hw, silu=True, out=None):
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    if out is None:
        out = torch.empty(n * hw, c0 + c1, device=x0.device, dtype=torch.float32)
    L.call("ldmk_gn_apply", _ptr(x0), c0, _ptr(x1), c1, _ptr(coef), _ptr(out), n, hw, 1 if silu else 0, stream())
    return out


def gn_apply_ps(x0, x1, coef, n, hw, h2_flag, silu=True, out=None):
    """ldmk_gn_apply with the result written once in the F16X2 form of the PS layout ([n hw][c0 + c1]): the A operand of a 3x3
    convolution on a conv-mode pre-split tile (conv3x3_ps)."""
    c0 = x0.shape[-1]
    c1 = 0 if x1 is None else x1.shape[-1]
    if out is None:
        out = ps_empty(n * hw, c0 + c1, 1, x0.device, h2=True)
    L.call("ldmk_gn_apply_ps_h2", _ptr(x0), c0, _ptr(x1), c1, _ptr(coef), _ptr(out), n, hw, 1 if silu else 0, _ptr(h2_flag), stream())
    return out


def conv3x3_ps(y_ps, n, h, w_, cin, wp, wps, range_flag, bias=None, stride=1, pad_lo=1, batch_vec=None, residual=None, out=None,
               tile_cfg=27, splitk=1, splitk_ws=None, stats_out=None):
    """3x3 convolution on a conv-mode pre-split tile (csrc/igemm_ps.hip: igemm_psc_kernel): y_ps = the (normalised) input
    [n h w][cin] in the F16X2 PS layout (gn_apply_ps / pack_ps), wp = pack_conv3x3 weights [9 cin][cout], wps = pack_wps(wp, h2=True)."""
    cout = wp.shape[1]
    oh = (h + 2 * pad_lo - 3) // stride + 1 if pad_lo == 1 else (h + 1 - 3) // stride + 1
    ow = (w_ + 2 * pad_lo - 3) // stride + 1 if pad_lo == 1 else (w_ + 1 - 3) // stride + 1
    if out is None:
        out = torch.empty(n, oh, ow, cout, device=wp.device, dtype=torch.float32)
    a = make_igemm_args(n * oh * ow, cout, 9 * cin, None, cin, wp, out, cout, oh * ow, conv=(h, w_, oh, ow, stride, pad_lo, 0), bias=bias,
                        batch_vec=batch_vec, batch_vec_ld=0 if batch_vec is None else batch_vec.stride(0), residual=residual,
                        tile_cfg=tile_cfg, splitk=splitk, splitk_ws=splitk_ws, a_ps=y_ps, w_ps=wps, range_flag=range_flag)
    if stats_out is not None:
        a.stats_out = stats_out.data_ptr()
        a._keep = a._keep + (stats_out,)
    igemm(a)
    return out


def ln_stats(x2d, eps=1e-5, out=None, split=None):
    """(mean, rstd) per row; split: a bf16 tensor [3][rows][ld] that receives the rows' exact three-way split (a_split operand)."""
    rows, c = x2d.shape
    if out is None:
        out = torch.empty(rows, 2, device=x2d.device, dtype=torch.float32)
    if split is not None:
        L.call("ldmk_ln_stats_split", _ptr(x2d), rows, c, float(eps), _ptr(out), _ptr(split), split.shape[-1], stream())
    else:
        L.call("ldmk_ln_stats", _ptr(x2d), rows, c, float(eps), _ptr(out), stream())
    return out


# ---- the PS layout (include/ldmk.h, csrc/igemm_ps.hip): operands of the pre-split bf16x3 tiles (tile_cfg 23..28) ------------
PS_TILE0 = 23                     # tile_cfg of the first pre-split tile (256x160); 24: 256x320, 25: 256x256, 26: 128x320, 27: 128x160, 28: 128x256


def ps_empty(rows, k, batch=1, device="cuda", h2=False):
    """Uninitialised PS-layout buffer(s) for a [rows][k] matrix: uint8 [batch][ldmk_ps_bytes] (h2: the two-plane fp16 form of the
    F16X2 arithmetic, ldmk_ps_bytes_h2)."""
    nb = (L.load().ldmk_ps_bytes_h2 if h2 else L.load().ldmk_ps_bytes)(int(rows), int(k))
    assert nb > 0, (rows, k)
    return torch.empty(batch, nb, device=device, dtype=torch.uint8)


F16X2_A_EXP = 6                   # LDMK_F16X2_A_EXP: activations are scaled by 2^6 in the F16X2 arithmetic


def pack_ps(x, out=None, h2_flag=None):
    """x: fp32 [rows][k] (or [batch][rows][k]) row-major on the GPU -> the PS layout of its exact three-way bf16 split; with h2_flag
    (device int32 range flag): the two fp16 planes of 2^6 x (F16X2)."""
    batch = x.shape[0] if x.dim() == 3 else 1
    rows, k = x.shape[-2], x.shape[-1]
    assert x.is_contiguous()
    if out is None:
        out = ps_empty(rows, k, batch, x.device, h2=h2_flag is not None)
    if h2_flag is not None:
        L.call("ldmk_pack_ps_h2", _ptr(x), rows, k, k, 1, batch, rows * k, F16X2_A_EXP, _ptr(h2_flag), _ptr(out), stream())
    else:
        L.call("ldmk_pack_ps", _ptr(x), rows, k, k, 1, batch, rows * k, _ptr(out), stream())
    return out


_WPS_H2_EXP = {}                  # data_ptr of an F16X2 PS weight image -> (weakref, scale exponent)


def h2_scale_exp(w):
    """The exponent e with max |2^e w| in [2^13, 2^14) (one host read of max |w|)."""
    import math
    mx = float(w.abs().max().item())
    e = 13 - math.floor(math.log2(mx)) if mx > 0.0 and math.isfinite(mx) else 0
    return max(-60, min(60, e))


def pack_wps(w, batch=1, h2=False):
    """Packed weights w[K][N] (pack_linear / pack_conv3x3 layout; or [batch][K][N]) -> PS layout of X[N][K] (row = output column):
    the w_ps operand of the pre-split tiles.  h2: the two fp16 planes of 2^e w, e per matrix (remembered for make_igemm_args)."""
    if batch > 1:
        assert w.dim() == 3 and w.shape[0] == batch and w.is_contiguous()
        K, N = w.shape[1], w.shape[2]
    else:
        assert w.dim() == 2 and w.is_contiguous()
        K, N = w.shape
    out = ps_empty(N, K, batch, w.device, h2=h2)
    if h2:
        e = h2_scale_exp(w)
        L.call("ldmk_pack_ps_h2", _ptr(w), N, K, 1, N, batch, K * N, e, 0, _ptr(out), stream())
        ptr = out.data_ptr()
        _WPS_H2_EXP[ptr] = (weakref.ref(out), e)
        weakref.finalize(out, _WPS_H2_EXP.pop, ptr, None)
    else:
        L.call("ldmk_pack_ps", _ptr(w), N, K, 1, N, batch, K * N, _ptr(out), stream())
    return out


def ln_stats_ps(x2d, eps=1e-5, out=None, ps=None, guard=0.0, flag=None, h2_flag=None):
    """LayerNorm (mean, rstd) per row AND the rows in the PS layout (one pass); returns (stats, ps).  h2_flag: the F16X2 planes."""
    rows, c = x2d.shape
    if out is None:
        out = torch.empty(rows, 2, device=x2d.device, dtype=torch.float32)
    if ps is None:
        ps = ps_empty(rows, c, 1, x2d.device, h2=h2_flag is not None)
    if h2_flag is not None:
        L.call("ldmk_ln_stats_ps_h2", _ptr(x2d), rows, c, float(eps), _ptr(out), _ptr(ps), float(guard), _ptr(flag), _ptr(h2_flag), stream())
    else:
        L.call("ldmk_ln_stats_ps", _ptr(x2d), rows, c, float(eps), _ptr(out), _ptr(ps), float(guard), _ptr(flag), stream())
    return out, ps


def unpack_ps_h2(ps, rows, k):
    """Test helper: F16X2 PS layout -> (hi, lo) float32 [rows][k] tensors of the SCALED operand."""
    nb = (rows + 31) // 32
    t = ps.reshape(-1).view(torch.float16).reshape(nb, k // 16, 2, 2, 32, 8)
    t = t.permute(2, 0, 4, 1, 3, 5).reshape(2, nb * 32, k)[:, :rows]
    return t[0].float(), t[1].float()


def unpack_ps(ps, rows, k):
    """Test helper: PS layout -> (hi, mid, lo) float32 [rows][k] tensors (host-side index arithmetic, any device)."""
    nb = (rows + 31) // 32
    t = ps.reshape(-1).view(torch.bfloat16).reshape(nb, k // 16, 3, 2, 32, 8)       # [row block][k slab][plane][k half][row][8 k]
    t = t.permute(2, 0, 4, 1, 3, 5).reshape(3, nb * 32, k)[:, :rows]
    return t[0].float(), t[1].float(), t[2].float()


def make_post_args(src, M, N, rows_per_sample, nslab=1, slab_stride=None, bias=None, batch_vec=None, batch_vec_ld=0,
                   residual=None, ldr=None, geglu=False, raw_out=None, ld_raw=None, norm=L.POST_NONE, x1=None, c1=0,
                   gamma=None, beta=None, eps=1e-5, silu=False, norm_out=None, ld_norm=None, groups=32, alpha=1.0):
    """ldmk_post_args (include/ldmk.h): the split-K reduce + GEMM epilogue + GroupNorm / LayerNorm launch."""
    a = L.PostArgs()
    a.src, a.nslab, a.slab_stride = _ptr(src), nslab, (M * N if slab_stride is None else slab_stride)
    a.M, a.N, a.rows_per_sample, a.alpha = M, N, rows_per_sample, alpha
    a.bias, a.batch_vec, a.batch_vec_ld = _ptr(bias), _ptr(batch_vec), batch_vec_ld
    a.residual, a.ldr = _ptr(residual), (N if ldr is None else ldr)
    a.geglu = 1 if geglu else 0
    a.raw_out, a.ld_raw = _ptr(raw_out), ((N // 2 if geglu else N) if ld_raw is None else ld_raw)
    a.norm, a.x1, a.c1, a.groups, a.eps = norm, _ptr(x1), c1, groups, float(eps)
    a.gamma, a.beta, a.silu = _ptr(gamma), _ptr(beta), 1 if silu else 0
    a.norm_out, a.ld_norm = _ptr(norm_out), ((N + c1) if ld_norm is None else ld_norm)
    return a


def post(args, device=None):
    """ldmk_post; the row-tiled GroupNorm form (large images) gets its scratch here (programs allocate it from their pool).
    `device`: where the operands live (default: the current CUDA device, which is the rank's own under torchrun)."""
    need = L.load().ldmk_post_scratch_elems(C.byref(args))
    if need > 0 and not args.gn_scratch:
        scratch = torch.empty(need, device=device if device is not None else torch.device("cuda", torch.cuda.current_device()),
                              dtype=torch.float32)
        args.gn_scratch, args.gn_scratch_elems = scratch.data_ptr(), need
        args._scratch = scratch
    L.call("ldmk_post", C.byref(args), stream())


# ------------------------------------------------------------------------------------------ igemm
def make_igemm_args(M, N, K, a0, c0, w, out, ldc, rows_per_sample, a1=None, c1=0, conv=None, tf=L.TF_NONE,
                    tf_coef=None, row_stats=None, ln_gamma=None, ln_beta=None, b_trans=False, ldb=None, bias=None,
                    batch_vec=None, batch_vec_ld=0, residual=None, epi=L.EPI_NONE, batch=1, a_bstride=0, w_bstride=0,
                    out_bstride=0, alpha=1.0, splitk=0, splitk_ws=None, w_frag=None, tile_cfg=0, compute=0, ln_colsum=None,
                    splitk_counters=None, raw_slabs=False, a_split=None, w_bf16t=None, a_ps=None, w_ps=None, out_ps=None,
                    range_flag=None, attn_kv=None):
    a = L.IgemmArgs()
    # the struct holds raw device pointers: keep every operand alive as long as the args object lives (a temporary passed
    # inline -- bias=b.cuda() -- would otherwise be freed, and its block possibly re-used, before the launch is enqueued)
    a._keep = (a0, a1, w, out, tf_coef, row_stats, ln_gamma, ln_beta, bias, batch_vec, residual, splitk_ws, w_frag, ln_colsum,
               splitk_counters, a_split, w_bf16t, a_ps, w_ps, out_ps)
    a.M, a.N, a.K = M, N, K
    a.a0, a.a1, a.c0, a.c1 = _ptr(a0), _ptr(a1), c0, c1
    if conv is not None:
        a.a_mode = L.A_CONV3X3
        a.in_h, a.in_w, a.out_h, a.out_w, a.stride, a.pad_lo, a.upsample = conv
    else:
        a.a_mode = L.A_ROWS
    a.a_tf = tf
    a.tf_coef, a.row_stats, a.ln_gamma, a.ln_beta = _ptr(tf_coef), _ptr(row_stats), _ptr(ln_gamma), _ptr(ln_beta)
    a.rows_per_sample = rows_per_sample
    a.w, a.b_trans = _ptr(w), 1 if b_trans else 0
    a.ldb = ldb if ldb is not None else (K if b_trans else N)
    a.bias, a.batch_vec, a.batch_vec_ld = _ptr(bias), _ptr(batch_vec), batch_vec_ld
    a.residual, a.epi = _ptr(residual), epi
    a.out, a.ldc, a.batch = _ptr(out), ldc, batch
    a.a_bstride, a.w_bstride, a.out_bstride = a_bstride, w_bstride, out_bstride
    a.alpha = alpha
    a.splitk = splitk
    a.w_frag, a.tile_cfg, a.compute = _ptr(w_frag), tile_cfg, compute
    a.ln_colsum = _ptr(ln_colsum)
    a.raw_slabs = 1 if raw_slabs else 0
    if splitk_counters is not None:
        a.splitk_counters, a.splitk_counters_len = splitk_counters.data_ptr(), splitk_counters.numel()
    if splitk_ws is not None:
        a.splitk_ws, a.splitk_ws_elems = splitk_ws.data_ptr(), splitk_ws.numel()
    if w_bf16t is not None:       # [N][ld] bf16 image of w (pack_wbf16t) for LDMK_COMPUTE_BF16
        a.w_split, a.w_split_ld = _ptr(w_bf16t), w_bf16t.shape[-1]
    if a_split is not None:       # [3][M][ld] bf16 images of a0 (ln_stats(..., split=...)); used by the bf16x3 LDS-tiled plans only
        a.a_split, a.a_split_ld = _ptr(a_split), a_split.shape[-1]
    if a_ps is not None:          # pre-split tiles (tile_cfg 23..33): both operands in the PS layout
        a.a_ps, a.w_ps, a.out_ps = _ptr(a_ps), _ptr(w_ps), _ptr(out_ps)
        if batch > 1:
            a.a_ps_bstride, a.w_ps_bstride = a_ps.shape[-1], w_ps.shape[-1]
        hit = _WPS_H2_EXP.get(w_ps.data_ptr())
        if hit is not None and hit[0]() is w_ps:        # F16X2 planes (pack_wps(h2=True)): a_ps / out_ps are in that form too
            a.compute, a.w_scale_exp, a.range_flag = L.COMPUTE_F16X2, hit[1], _ptr(range_flag)
            a._keep = a._keep + (range_flag,)
            if attn_kv is not None:                     # (kv tile buffer, tokens per sample, heads): the fused QKV projection writes
                a.attn_kv_out, a.attn_tokens, a.attn_heads = _ptr(attn_kv[0]), int(attn_kv[1]), int(attn_kv[2])     # the attention's K / V tiles
                a._keep = a._keep + (attn_kv[0],)
        else:
            a.compute = L.COMPUTE_BF16X3
        return a
    if compute == L.COMPUTE_BF16X3 and not set_split(a):
        raise ValueError("COMPUTE_BF16X3: no split images registered for this weight (ops.pack_wsplit) or b_trans set")
    if compute == L.COMPUTE_F16X2 and not set_split_h2(a, range_flag):
        raise ValueError("COMPUTE_F16X2: no fp16 images registered for this weight (ops.pack_wsplit_h2), b_trans set, or no range_flag")
    return a


def igemm(args):
    L.call("ldmk_igemm", C.byref(args), stream())


# ---- LDMK_COMPUTE_BF16X3: the weights as three bf16 images of their exact split (include/ldmk.h) -------------------
_SPLIT = {}          # data_ptr of a packed fp32 weight [K][N] (or a batch of them) -> (bf16 tensor, ld, batch stride in elements)


def pack_wsplit(w, batch=1):
    """w: [K][N] fp32 (or [batch][K][N]) on the GPU -> bf16 [batch][3][N][ld] (ldmk_pack_wsplit), registered under w's address
    so that make_igemm_args(..., compute=COMPUTE_BF16X3) finds it."""
    if batch > 1:
        assert w.dim() == 3 and w.shape[0] == batch and w.is_contiguous()
        K, N = w.shape[1], w.shape[2]
    else:
        assert w.dim() == 2 and w.is_contiguous()
        K, N = w.shape
    ld = (K + 7) // 8 * 8
    out = torch.empty(batch, 3, N, ld, device=w.device, dtype=torch.bfloat16)
    L.call("ldmk_pack_wsplit", _ptr(w), K, N, N, batch, K * N, _ptr(out), ld, stream())
    ptr = w.data_ptr()
    _SPLIT[ptr] = (out, ld, 3 * N * ld, weakref.ref(w), w._version)
    # the images die with the weight they were made from (a deleted model must give back its ~6 bytes per parameter); the
    # entry is dropped only if it still is THIS weight's (the allocator may have handed the address to a newer weight).
    # Packed weights must not be written through raw pointers after this call: only in-place torch writes move _version.
    weakref.finalize(w, _drop_split, ptr, out.data_ptr())
    return out


def _drop_split(ptr, img_ptr):
    hit = _SPLIT.get(ptr)
    if hit is not None and hit[0].data_ptr() == img_ptr:
        del _SPLIT[ptr]


def pack_wbf16t(w):
    """w: [K][N] fp32 on the GPU -> bf16 [N][ld] (rounded to nearest even, transposed, K-contiguous; ldmk_pack_wbf16t): the
    w_split operand of an LDMK_COMPUTE_BF16 GEMM (the training step's forward weights, train.repack_bf16_weights)."""
    assert w.dim() == 2 and w.is_contiguous()
    K, N = w.shape
    ld = (K + 7) // 8 * 8
    out = torch.empty(N, ld, device=w.device, dtype=torch.bfloat16)
    L.call("ldmk_pack_wbf16t", _ptr(w), K, N, N, _ptr(out), ld, stream())
    return out


def split_of(w_ptr):
    """The registered split images of the weight at this address, or None (also when the weight tensor has since been freed
    or written in place: its address / contents may no longer be what was split)."""
    hit = _SPLIT.get(w_ptr)
    if hit is None:
        return None
    src = hit[3]()
    if src is None or src.data_ptr() != w_ptr or src._version != hit[4]:
        del _SPLIT[w_ptr]
        return None
    return hit


# ---- LDMK_COMPUTE_F16X2: the weights as two fp16 images of 2^e w (include/ldmk.h) ---------------------------------------------
_SPLIT_H2 = {}       # data_ptr of a packed fp32 weight -> (f16 tensor, ld, batch stride in elements, weakref, version, scale exponent)


def pack_wsplit_h2(w, batch=1):
    """w: [K][N] fp32 (or [batch][K][N]) on the GPU -> fp16 [batch][2][N][ld] (ldmk_pack_wsplit_h2): the images hi, lo of 2^e w
    with e chosen so that max |2^e w| lies in [2^13, 2^14) (one host read of max |w| at pack time); registered under w's address."""
    if batch > 1:
        assert w.dim() == 3 and w.shape[0] == batch and w.is_contiguous()
        K, N = w.shape[1], w.shape[2]
    else:
        assert w.dim() == 2 and w.is_contiguous()
        K, N = w.shape
    mx = float(w.abs().max().item())
    import math
    e = 13 - math.floor(math.log2(mx)) if mx > 0.0 and math.isfinite(mx) else 0
    e = max(-60, min(60, e))
    ld = (K + 7) // 8 * 8
    out = torch.empty(batch, 2, N, ld, device=w.device, dtype=torch.float16)
    L.call("ldmk_pack_wsplit_h2", _ptr(w), K, N, N, batch, K * N, e, _ptr(out), ld, stream())
    ptr = w.data_ptr()
    _SPLIT_H2[ptr] = (out, ld, 2 * N * ld, weakref.ref(w), w._version, e)
    weakref.finalize(w, _drop_split_h2, ptr, out.data_ptr())
    return out


def _drop_split_h2(ptr, img_ptr):
    hit = _SPLIT_H2.get(ptr)
    if hit is not None and hit[0].data_ptr() == img_ptr:
        del _SPLIT_H2[ptr]


def split_h2_of(w_ptr):
    hit = _SPLIT_H2.get(w_ptr)
    if hit is None:
        return None
    src = hit[3]()
    if src is None or src.data_ptr() != w_ptr or src._version != hit[4]:
        del _SPLIT_H2[w_ptr]
        return None
    return hit


def set_split_h2(a, range_flag, w_ptr=None):
    """Switch igemm args to the F16X2 arithmetic when the weight's fp16 images exist; returns True if it did."""
    hit = split_h2_of(a.w if w_ptr is None else w_ptr)
    if hit is None or a.b_trans or range_flag is None:
        return False
    a.w_split, a.w_split_ld, a.w_split_bstride = hit[0].data_ptr(), hit[1], hit[2]
    a.w_scale_exp, a.range_flag = hit[5], range_flag.data_ptr()
    a.compute = L.COMPUTE_F16X2
    a._keep = getattr(a, "_keep", ()) + (range_flag,)
    return True


def set_split(a, w_ptr=None):
    """Switch igemm args to the fp32-accurate bf16x3 arithmetic when the weight's split images exist; returns True if it did."""
    hit = split_of(a.w if w_ptr is None else w_ptr)
    if hit is None or a.b_trans:
        return False
    a.w_split, a.w_split_ld, a.w_split_bstride = hit[0].data_ptr(), hit[1], hit[2]
    a.compute = L.COMPUTE_BF16X3
    return True


def conv3x3(x, wp, bias=None, x1=None, stride=1, pad_lo=1, upsample=False, coef=None, silu=True, batch_vec=None,
            residual=None, out=None, out_hw=None, compute=0, range_flag=None):
    """x: (n,h,w,c0) [+ x1 (n,h,w,c1) channel-concat]; wp: [9*(c0+c1)][cout] -> (n,oh,ow,cout)."""
    n, h, w_, c0 = x.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    cout = wp.shape[1]
    if out_hw is None:
        if upsample:
            oh, ow = 2 * h, 2 * w_
        else:
            oh = (h + 2 * pad_lo - 3) // stride + 1 if pad_lo == 1 else (h + 1 - 3) // stride + 1
            ow = (w_ + 2 * pad_lo - 3) // stride + 1 if pad_lo == 1 else (w_ + 1 - 3) // stride + 1
    else:
        oh, ow = out_hw
    if out is None:
        out = torch.empty(n, oh, ow, cout, device=x.device, dtype=torch.float32)
    tf = L.TF_NONE if coef is None else (L.TF_AFFINE_SILU if silu else L.TF_AFFINE)
    a = make_igemm_args(n * oh * ow, cout, 9 * (c0 + c1), x, c0, wp, out, cout, oh * ow, a1=x1, c1=c1,
                        conv=(h, w_, oh, ow, stride, pad_lo, 1 if upsample else 0), tf=tf, tf_coef=coef, bias=bias,
                        batch_vec=batch_vec, batch_vec_ld=0 if batch_vec is None else batch_vec.stride(0),
                        residual=residual, compute=compute, range_flag=range_flag)
    igemm(a)
    return out


def linear(x2d, wp, bias=None, x1=None, rows_per_sample=None, coef=None, silu=False, row_stats=None, ln_gamma=None,
           ln_beta=None, batch_vec=None, residual=None, geglu=False, out=None, b_trans=False, w_frag=None, tile_cfg=0,
           stats_out=None, compute=0, ln_colsum=None, a_split=None, range_flag=None):
    """x2d: [M][c0] (+ x1 [M][c1]); wp: [K][N] (or torch [N][K] with b_trans) -> [M][N] (N/2 for geglu)."""
    M, c0 = x2d.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    K = c0 + c1
    N = wp.shape[0] if b_trans else wp.shape[1]
    ncol = N // 2 if geglu else N
    if out is None:
        out = torch.empty(M, ncol, device=x2d.device, dtype=torch.float32)
    tf = L.TF_NONE
    if coef is not None:
        tf = L.TF_AFFINE_SILU if silu else L.TF_AFFINE
    elif row_stats is not None:      # ln_colsum: LayerNorm folded through the product (wp, bias, ln_colsum from fold_layernorm)
        tf = L.TF_LAYERNORM if ln_colsum is None else L.TF_LAYERNORM_FOLDED
    a = make_igemm_args(M, N, K, x2d, c0, wp, out, ncol, rows_per_sample or M, a1=x1, c1=c1, tf=tf, tf_coef=coef,
                        row_stats=row_stats, ln_gamma=ln_gamma, ln_beta=ln_beta, b_trans=b_trans, bias=bias,
                        batch_vec=batch_vec, batch_vec_ld=0 if batch_vec is None else batch_vec.stride(0),
                        residual=residual, epi=L.EPI_GEGLU if geglu else L.EPI_NONE, w_frag=w_frag, tile_cfg=tile_cfg,
                        compute=compute, ln_colsum=ln_colsum, a_split=a_split, range_flag=range_flag)
    if stats_out is not None:
        a.stats_out = _ptr(stats_out)
    igemm(a)
    return out


def fold_layernorm(wp, gamma, beta, bias=None):
    """Packed Linear weight wp [K][N] + the LayerNorm in front of it -> (diag(gamma) wp, colsum [N], beta^T wp + bias [N]):
    the operands of LDMK_TF_LAYERNORM_FOLDED (ldmk.h)."""
    K, N = wp.shape
    w2 = torch.empty(K, N, device=wp.device, dtype=torch.float32)
    cs = torch.empty(N, device=wp.device, dtype=torch.float32)
    b2 = torch.empty(N, device=wp.device, dtype=torch.float32)
    L.call("ldmk_fold_layernorm", _ptr(wp), wp.stride(0), K, N, _ptr(gamma), _ptr(beta), _ptr(bias), _ptr(w2), _ptr(cs), _ptr(b2),
           stream())
    return w2, cs, b2


# ------------------------------------------------------------------------------------------ Winograd F(2x2, 3x3)
_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def pack_winograd(w):
    """torch conv weight [cout][cin][3][3] -> U[16][cin][cout] with U[4i+j] = (G g G^T)[i][j] (computed in float64): the
    16 [K][N] weight matrices of the batched GEMM between the Winograd transforms."""
    _chk(w, "pack_winograd")
    g = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    u = torch.einsum("ia,ocab,jb->ijco", g, w.double(), g)
    return u.reshape(16, w.shape[1], w.shape[0]).float().contiguous()


def conv3x3_winograd(x, u, bias=None, x1=None, coef=None, silu=True, batch_vec=None, residual=None, out=None, stats_out=None,
                     scratch=None):
    """Stride-1 pad-1 3x3 convolution through Winograd F(2x2,3x3): x (n,h,w,c0) [+ x1 channel concat], u = pack_winograd(w).
    `coef`: GroupNorm scale/shift planes applied (with SiLU if `silu`) to the input first."""
    n, h, w_, c0 = x.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    K, N = u.shape[1], u.shape[2]
    assert K == c0 + c1
    tiles = n * (h // 2) * (w_ // 2)
    if scratch is None:
        scratch = (torch.empty(16, tiles, K, device=x.device), torch.empty(16, tiles, N, device=x.device))
    V, Mb = scratch
    if out is None:
        out = torch.empty(n, h, w_, N, device=x.device, dtype=torch.float32)
    L.call("ldmk_winograd_input", _ptr(x), c0, _ptr(x1), c1, _ptr(coef), 1 if (silu and coef is not None) else 0, n, h, w_,
           _ptr(V), stream())
    a = make_igemm_args(tiles, N, K, V, K, u, Mb, N, tiles, batch=16, a_bstride=tiles * K, w_bstride=K * N,
                        out_bstride=tiles * N)
    igemm(a)
    L.call("ldmk_winograd_output", _ptr(Mb), _ptr(bias), _ptr(batch_vec), 0 if batch_vec is None else batch_vec.stride(0),
           _ptr(residual), _ptr(out), _ptr(stats_out), n, h, w_, N, stream())
    return out


_UP_TAPS = (((0,), (1, 2)), ((0, 1), (2,)))      # [parity][collapsed tap] -> the 3x3 taps it sums


def pack_upconv(w):
    """torch conv weight [cout][cin][3][3] of an Upsample layer -> Wp[4 phases][4 taps * cin][cout]: for output parity (a, b)
    the 2x2-tap kernel whose tap (i, j) sums the 3x3 taps that read the same low-resolution pixel (float64 sums)."""
    _chk(w, "pack_upconv")
    cout, cin = w.shape[0], w.shape[1]
    wd = w.double()
    out = torch.empty(4, 4 * cin, cout, dtype=torch.float64, device=w.device)
    for a in range(2):
        for b in range(2):
            for i in range(2):
                for j in range(2):
                    acc = sum(wd[:, :, dy, dx] for dy in _UP_TAPS[a][i] for dx in _UP_TAPS[b][j])        # [cout][cin]
                    out[2 * a + b, (2 * i + j) * cin:(2 * i + j + 1) * cin] = acc.t()
    return out.float().contiguous()


def upsample_conv3x3_phases(x, wp4, bias=None, out=None, stats_out=None, scratch=None):
    """Nearest-x2 upsample + 3x3 conv through the four 2x2-tap phase convolutions: x (n,h,w,c) -> (n,2h,2w,cout)."""
    n, h, w_, c = x.shape
    N = wp4.shape[2]
    pix = n * h * w_
    if scratch is None:
        scratch = (torch.empty(4, pix, 4 * c, device=x.device), torch.empty(4, pix, N, device=x.device))
    A, Pm = scratch
    if out is None:
        out = torch.empty(n, 2 * h, 2 * w_, N, device=x.device, dtype=torch.float32)
    L.call("ldmk_upconv_gather", _ptr(x), c, n, h, w_, _ptr(A), stream())
    a = make_igemm_args(pix, N, 4 * c, A, 4 * c, wp4, Pm, N, pix, batch=4, a_bstride=pix * 4 * c, w_bstride=4 * c * N,
                        out_bstride=pix * N)
    igemm(a)
    L.call("ldmk_upconv_scatter", _ptr(Pm), _ptr(bias), _ptr(out), _ptr(stats_out), n, h, w_, N, stream())
    return out


def bmm(a, b, b_trans, alpha=1.0, out=None):
    """Batched a[B][M][K] x (b[B][K][N] | b[B][N][K]^T) -> [B][M][N] on the matrix cores."""
    B, M, K = a.shape
    N = b.shape[1] if b_trans else b.shape[2]
    if out is None:
        out = torch.empty(B, M, N, device=a.device, dtype=torch.float32)
    args = make_igemm_args(M, N, K, a, K, b, out, N, M, b_trans=b_trans, batch=B, a_bstride=M * K,
                           w_bstride=b.shape[1] * b.shape[2], out_bstride=M * N, alpha=alpha)
    igemm(args)
    return out


# ------------------------------------------------------------------------------------------ attention
def attn_self(qkv, n, tokens, heads, out=None, x3=False, presplit=False, h2_flag=None):
    """x3: both products in the fp32-accurate three-way bf16 split arithmetic (ldmk_attn_self_x3); presplit: its form with K / V
    split once by a pre-pass and moved to LDS by LDS-DMA (ldmk_attn_self_x3p: bitwise the same result); h2_flag (device int32
    tensor): the two-way fp16 split with three products per term (ldmk_attn_self_h2), the flag raised when an operand leaves
    its range."""
    C_ = heads * 32
    if out is None:
        out = torch.empty(n * tokens, C_, device=qkv.device, dtype=torch.float32)
    if h2_flag is not None:
        kv = torch.empty(L.load().ldmk_attn_kv_split_h2_bytes(n, tokens, heads), device=qkv.device, dtype=torch.uint8)
        L.call("ldmk_attn_self_h2", _ptr(qkv), _ptr(kv), _ptr(out), _ptr(h2_flag), n, tokens, heads, 32 ** -0.5, stream())
        return out
    if presplit:
        kv = torch.empty(L.load().ldmk_attn_kv_split_bytes(n, tokens, heads), device=qkv.device, dtype=torch.uint8)
        L.call("ldmk_attn_self_x3p", _ptr(qkv), _ptr(kv), _ptr(out), n, tokens, heads, 32 ** -0.5, stream())
        return out
    L.call("ldmk_attn_self_x3" if x3 else "ldmk_attn_self", _ptr(qkv), _ptr(out), n, tokens, heads, 32 ** -0.5, stream())
    return out


def attn_cross(q, k, v, n, tokens, ctx_len, heads, out=None):
    C_ = heads * 32
    if out is None:
        out = torch.empty(n * tokens, C_, device=q.device, dtype=torch.float32)
    L.call("ldmk_attn_cross", _ptr(q), q.stride(0), _ptr(k), _ptr(v), k.stride(0), _ptr(out), out.stride(0), n, tokens,
           ctx_len, heads, 32 ** -0.5, stream())
    return out


def softmax_rows_(x2d, scale):
    rows, cols = x2d.shape
    L.call("ldmk_softmax_rows", _ptr(x2d), rows, cols, float(scale), stream())
    return x2d


# ------------------------------------------------------------------------------------------ small ops
def dense_small(x, wp, bias=None, silu_in=False, out=None):
    rows, K = x.shape
    N = wp.shape[1]
    if out is None:
        out = torch.empty(rows, N, device=x.device, dtype=torch.float32)
    L.call("ldmk_dense_small", _ptr(x), x.stride(0), _ptr(wp), _ptr(bias), _ptr(out), out.stride(0), rows, K, N,
           1 if silu_in else 0, stream())
    return out


def timestep_freqs(dim, max_period=10000, device="cuda"):
    """Frequency table of timestep_embedding (a per-model constant; the sinusoid formula of
    "Attention Is All You Need" in the fp32 operation order the reference uses, util.py:161-163)."""
    import numpy as np
    half = dim // 2
    # exponent in the reference's fp32 operation order; exp itself in float64 then rounded once, so the
    # table is identical on every host (torch's CPU expf differs by 1 ulp between SIMD code paths)
    e = (np.float32(-math.log(max_period)) * np.arange(half, dtype=np.float32)) / np.float32(half)
    f = np.exp(e.astype(np.float64)).astype(np.float32)
    return torch.from_numpy(f).to(device)


def timestep_embedding(t, freqs, dim, out=None):
    n = t.shape[0]
    if out is None:
        out = torch.empty(n, dim, device=t.device, dtype=torch.float32)
    L.call("ldmk_timestep_embedding", _ptr(t), _ptr(freqs), _ptr(out), n, dim, stream())
    return out


def conv3x3_in(x0, wp, bias, cout, x1=None, out=None):
    """NCHW narrow input(s) -> NHWC (n,h,w,cout).  wp: [9*cin][cout]."""
    n, c0, h, w_ = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    if out is None:
        out = torch.empty(n, h, w_, cout, device=x0.device, dtype=torch.float32)
    L.call("ldmk_conv3x3_in", _ptr(x0), c0, _ptr(x1), c1, _ptr(wp), _ptr(bias), _ptr(out), n, h, w_, cout, stream())
    return out


def conv3x3_out(x, coef, wp, bias, cout, out=None, small=False):
    """NHWC (n,h,w,cin) -> GN+SiLU -> conv -> NCHW (n,cout,h,w).  wp: [9*cin][cout].  small: the 4x4-pixel tiling for
    small images (ldmk_conv3x3_out_small)."""
    n, h, w_, cin = x.shape
    if out is None:
        out = torch.empty(n, cout, h, w_, device=x.device, dtype=torch.float32)
    L.call("ldmk_conv3x3_out_small" if small else "ldmk_conv3x3_out", _ptr(x), _ptr(coef), _ptr(wp), _ptr(bias), _ptr(out),
           n, h, w_, cin, cout, stream())
    return out


def conv1x1_nchw(x, w, bias, out=None):
    n, cin, h, w_ = x.shape
    cout = w.shape[0]
    if out is None:
        out = torch.empty(n, cout, h, w_, device=x.device, dtype=torch.float32)
    L.call("ldmk_conv1x1_nchw", _ptr(x), _ptr(w), _ptr(bias), _ptr(out), n, h * w_, cin, cout, stream())
    return out


def vq_nearest(z, codebook, zq=None, idx=None):
    n, dim, h, w_ = z.shape
    if zq is None:
        zq = torch.empty_like(z)
    if idx is None:
        idx = torch.empty(n * h * w_, device=z.device, dtype=torch.int32)
    L.call("ldmk_vq_nearest", _ptr(z), _ptr(codebook), _ptr(zq), _ptr(idx), n, h * w_, dim, codebook.shape[0], stream())
    return zq, idx


def postprocess_frames(x, out=None):
    n, c, h, w_ = x.shape
    if out is None:
        out = torch.empty(n, h, w_, c, device=x.device, dtype=torch.float32)
    L.call("ldmk_postprocess_frames", _ptr(x), _ptr(out), n, c, h * w_, stream())
    return out


def add_rowvec_(x2d, vec, rows_per_sample):
    rows, c = x2d.shape
    L.call("ldmk_add_rowvec", _ptr(x2d), _ptr(vec), vec.stride(0), rows, c, rows_per_sample, stream())
    return x2d
