"""GPU: the small-batch (latency-bound) route -- the slab GEMM (csrc/sgemm.hip, tile_cfg 13..16) and ldmk_post
(csrc/post.hip) -- against float64 PyTorch references of the same ops, through the C ABI.

fp32 products summed in a different order than PyTorch: 1e-4 / 1e-4 like the other GEMM tests (measured: see DESIGN)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rnd

pytestmark = pytest.mark.gpu
SLAB_TILES = {13: (2, 1, 4), 14: (2, 2, 4), 15: (1, 1, 4), 16: (1, 2, 4), 17: (1, 1, 8), 18: (1, 1, 16), 19: (2, 1, 8), 20: (1, 2, 8)}


@pytest.fixture(scope="module")
def ops():
    from dsml_thesis_amd import ops as ops_
    from dsml_thesis_amd import lib
    lib.load()
    return ops_


def close(a, b, rtol=1e-4, atol=1e-4):
    torch.testing.assert_close(a.float().cpu(), torch.as_tensor(b).float().cpu(), rtol=rtol, atol=atol)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(y):
    return y.permute(0, 3, 1, 2).contiguous().cpu()


def _run(ops, a, cfg, splitk, M, N, raw=False):
    a.tile_cfg, a.splitk = cfg, splitk
    ws = torch.full((max(1, splitk) * M * N + 8,), float("nan"), device="cuda")
    a.splitk_ws, a.splitk_ws_elems = ws.data_ptr(), ws.numel()
    a.raw_slabs = 1 if raw else 0
    ops.igemm(a)
    return ws


@pytest.mark.parametrize("cfg", sorted(SLAB_TILES))
@pytest.mark.parametrize("splitk", [1, 2, 5])
@pytest.mark.parametrize("case", [(1, 640, 640, 8, 8), (1, 160, 320, 16, 16), (2, 64, 96, 5, 7), (1, 1280, 640, 8, 8)])
def test_slab_gemm_conv3x3(ops, cfg, splitk, case):
    """openaimodel.py:204,230: the ResBlock 3x3 convolutions at batch 1-2, every wave tile and K split, with the full
    epilogue (bias, timestep vector, residual, GroupNorm records); ragged image sizes (partial row tiles) included."""
    from dsml_thesis_amd import lib as L
    n, cin, cout, h, w = case
    x, wt, b = rnd(10, n, cin, h, w), rnd(11, cout, cin, 3, 3) / np.sqrt(9 * cin), 0.1 * rnd(12, cout)
    vec, res = rnd(13, n, cout), rnd(14, n, cout, h, w)
    use_vec = (h * w) % 64 == 0                    # a per-sample vector needs whole row tiles per sample
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1) + res.double()
    if use_vec:
        ref = ref + vec.double()[:, :, None, None]
    xd, wp, bd, vd, rd = nhwc(x), ops.pack_conv3x3(wt.cuda()), b.cuda(), (vec.cuda() if use_vec else None), nhwc(res)
    wf = ops.pack_wfrag(wp)
    M = n * h * w
    out = torch.empty(n, h, w, cout, device="cuda")
    stats = M % 32 == 0 and (h * w) % 32 == 0
    part = torch.zeros(max(1, M // 32), cout, 3, device="cuda")
    a = ops.make_igemm_args(M, cout, 9 * cin, xd, cin, wp, out, cout, h * w, conv=(h, w, h, w, 1, 1, 0), bias=bd,
                            batch_vec=vd, batch_vec_ld=cout, residual=rd, w_frag=wf)
    if stats:
        a.stats_out = part.data_ptr()
    tm, tn, nw = SLAB_TILES[cfg]
    if cout % (32 * tn) or nw * splitk > 9 * cin // 8:
        with pytest.raises(L.LdmkError, match="slab GEMM"):
            _run(ops, a, cfg, splitk, M, cout)
        return
    _run(ops, a, cfg, splitk, M, cout)
    close(nchw(out), ref.float())
    out2 = torch.empty_like(out)
    a.out = out2.data_ptr()
    _run(ops, a, cfg, splitk, M, cout)
    assert torch.equal(out, out2), "fixed summation order: bitwise reproducible"
    if stats:      # the GroupNorm partial records describe the stored tensor: (shift, sum of x - shift, sum of squares)
        o = out.reshape(M // 32, 32, cout).double()
        p = part.double()
        torch.testing.assert_close(p[..., 1] + 32 * p[..., 0], o.sum(1), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(p[..., 2] + 2 * p[..., 0] * p[..., 1] + 32 * p[..., 0] ** 2, (o * o).sum(1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("cfg", sorted(SLAB_TILES))
@pytest.mark.parametrize("splitk", [1, 3])
@pytest.mark.parametrize("case", [(64, 640, 640, "res"), (256, 320, 960, "none"), (1024, 160, 160, "affine"),
                                  (64, 640, 5120, "geglu"), (100, 160, 320, "concat"), (64, 640, 1920, "lnf")])
def test_slab_gemm_rows(ops, cfg, splitk, case):
    """Token-row Linear / 1x1 layers of the transformer blocks at batch 1 (attention.py:161-168,37-64,232-248;
    openaimodel.py:241): residual + per-sample vector, GroupNorm-affine prologue (proj_in), GEGLU, two-source concat (skip
    1x1), LayerNorm folded through the product -- against float64."""
    from dsml_thesis_amd import lib as L
    M, K, N, kind = case
    x = rnd(20, M, K) * 1.1 + 0.2
    w, b = rnd(21, N, K) / np.sqrt(K), 0.1 * rnd(22, N)
    wp, bp = ops.pack_linear(w.cuda()), b.cuda()
    kw, hw = {}, 64 if M % 64 == 0 else M
    xr = x.double()
    keep = []
    if kind == "res":
        res, vec = rnd(23, M, N), rnd(24, M // hw, N)
        ref = F.linear(xr, w.double(), b.double()) + res.double() + vec.double().repeat_interleave(hw, 0)
        kw.update(residual=res.cuda(), batch_vec=vec.cuda(), batch_vec_ld=N)
    elif kind == "affine":
        coef = torch.stack([1 + 0.2 * rnd(25, M // hw, K), 0.3 * rnd(26, M // hw, K)], 1)          # [n][2][K]
        ref = F.linear(xr * coef[:, 0].double().repeat_interleave(hw, 0) + coef[:, 1].double().repeat_interleave(hw, 0),
                       w.double(), b.double())
        kw.update(tf=L.TF_AFFINE, tf_coef=coef.cuda())
    elif kind == "geglu":
        wp, bp = ops.pack_geglu(w.cuda(), b.cuda())
        v_, g_ = F.linear(xr, w.double(), b.double()).chunk(2, dim=1)
        ref = v_ * F.gelu(g_)
        kw.update(epi=L.EPI_GEGLU)
    elif kind == "lnf":
        g, be = 1 + 0.2 * rnd(27, K), 0.2 * rnd(28, K)
        ref = F.linear(F.layer_norm(xr, (K,), g.double(), be.double(), 1e-5), w.double(), b.double())
        st = ops.ln_stats(x.cuda())
        wp, cs, bp = ops.fold_layernorm(wp, g.cuda(), be.cuda(), bp)
        kw.update(tf=L.TF_LAYERNORM_FOLDED, row_stats=st, ln_colsum=cs)
        keep += [st, cs]
    else:
        ref = F.linear(xr, w.double(), b.double())
    wf = ops.pack_wfrag(wp)
    ncol = N // 2 if kind == "geglu" else N
    out = torch.empty(M, ncol, device="cuda")
    xc = x.cuda()
    if kind == "concat":
        c0 = 96
        x0, x1 = xc[:, :c0].contiguous(), xc[:, c0:].contiguous()
        a = ops.make_igemm_args(M, N, K, x0, c0, wp, out, ncol, hw, a1=x1, c1=K - c0, bias=bp, w_frag=wf, **kw)
    else:
        a = ops.make_igemm_args(M, N, K, xc, K, wp, out, ncol, hw, bias=bp, w_frag=wf, **kw)
    tm, tn, nw = SLAB_TILES[cfg]
    bad = (N % (32 * tn) or (kind == "geglu" and (tn % 2 or splitk > 1)) or (kind in ("res", "affine") and hw % (32 * tm))
           or nw * splitk > K // 8)
    if bad:
        with pytest.raises(L.LdmkError, match="slab GEMM"):
            _run(ops, a, cfg, splitk, M, N)
        return
    _run(ops, a, cfg, splitk, M, N)
    close(out, ref.float())


def _slabs(ops, M, K, N, splitk, seed=30, geglu=False):
    """raw split-K slabs of x W from the slab GEMM (raw_slabs): what ldmk_post consumes."""
    from dsml_thesis_amd import lib as L
    x, w = rnd(seed, M, K), rnd(seed + 1, N, K) / np.sqrt(K)
    if geglu:
        wp, _ = ops.pack_geglu(w.cuda(), torch.zeros(N, device="cuda"))
    else:
        wp = ops.pack_linear(w.cuda())
    wf = ops.pack_wfrag(wp)
    xc = x.cuda()
    dummy = torch.empty(8, device="cuda")
    a = ops.make_igemm_args(M, N, K, xc, K, wp, dummy, N, M, w_frag=wf, epi=L.EPI_GEGLU if geglu else L.EPI_NONE)
    ws = _run(ops, a, 13 if not geglu else 14, splitk, M, N, raw=True)
    return ws, F.linear(x.double(), w.double())


@pytest.mark.parametrize("case", [(1, 64, 640, 0, 4, True), (1, 1024, 160, 0, 1, True), (2, 256, 320, 160, 3, True),
                                  (1, 64, 640, 640, 16, True), (1, 4096, 160, 320, 2, False), (1, 40, 64, 32, 2, False)])
def test_post_groupnorm(ops, case):
    """Split-K reduce + bias + timestep vector + residual, then GroupNorm(32)[+SiLU] over the channel concat with a skip
    tensor (groups straddle the seam: 320 + 160 -> 15 channels per group): openaimodel.py:201-203,264-275,736."""
    from dsml_thesis_amd import lib as L
    n, hw, N, c1, nslab, silu = case
    M = n * hw
    K = 256 if nslab <= 8 else 1024
    if nslab > 1:
        ws, prod = _slabs(ops, M, K, N, nslab)
    else:
        prod = rnd(40, M, N).double()
        ws = prod.float().cuda()
    b, vec, res = 0.1 * rnd(41, N), rnd(42, n, N), rnd(43, M, N)
    x1 = rnd(44, M, c1) * 1.3 + 0.4 if c1 else None
    C = N + c1
    gamma, beta = 1 + 0.1 * rnd(45, C), 0.1 * rnd(46, C)
    raw_ref = prod + b.double() + vec.double().repeat_interleave(hw, 0) + res.double()
    cat = raw_ref if x1 is None else torch.cat([raw_ref, x1.double()], 1)
    y = F.group_norm(cat.view(n, hw, C).permute(0, 2, 1), 32, gamma.double(), beta.double(), 1e-5)
    y = (F.silu(y) if silu else y).permute(0, 2, 1).reshape(M, C)
    raw, out = torch.empty(M, N, device="cuda"), torch.empty(M, C, device="cuda")
    keep = (b.cuda(), vec.cuda(), res.cuda(), None if x1 is None else x1.cuda(), gamma.cuda(), beta.cuda())
    a = ops.make_post_args(ws, M, N, hw, nslab=nslab, bias=keep[0], batch_vec=keep[1], batch_vec_ld=N, residual=keep[2],
                           raw_out=raw, norm=L.POST_GROUPNORM, x1=keep[3], c1=c1, gamma=keep[4], beta=keep[5], eps=1e-5,
                           silu=silu, norm_out=out)
    ops.post(a)
    close(raw, raw_ref.float(), 1e-4, 1e-4)
    close(out, y.float(), 1e-4, 1e-4)
    out2 = torch.empty_like(out)
    a.norm_out = out2.data_ptr()
    ops.post(a)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("case", [(64, 640, 6), (1024, 160, 1), (256, 320, 3), (37, 1280, 2)])
def test_post_layernorm(ops, case):
    """reduce + bias + per-sample cross-attention vector + residual, then LayerNorm * gamma + beta per row: the residual
    stream and the normalised GEMM input of the next Linear in one launch (attention.py:203-205,211-215)."""
    from dsml_thesis_amd import lib as L
    M, N, nslab = case
    hw = M
    if nslab > 1:
        ws, prod = _slabs(ops, M, 256, N, nslab)
    else:
        prod = rnd(40, M, N).double()
        ws = prod.float().cuda()
    b, vec, res = 0.1 * rnd(41, N), rnd(42, 1, N), rnd(43, M, N) + 0.5
    g, be = 1 + 0.2 * rnd(47, N), 0.2 * rnd(48, N)
    raw_ref = prod + b.double() + vec.double() + res.double()
    y = F.layer_norm(raw_ref, (N,), g.double(), be.double(), 1e-5)
    raw, out = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    keep = (b.cuda(), vec.cuda(), res.cuda(), g.cuda(), be.cuda())
    a = ops.make_post_args(ws, M, N, hw, nslab=nslab, bias=keep[0], batch_vec=keep[1], batch_vec_ld=N, residual=keep[2],
                           raw_out=raw, norm=L.POST_LAYERNORM, gamma=keep[3], beta=keep[4], eps=1e-5, norm_out=out)
    ops.post(a)
    close(raw, raw_ref.float())
    close(out, y.float())


@pytest.mark.parametrize("case", [(64, 5120, 4), (256, 2560, 2), (1024, 1280, 1)])
def test_post_geglu_and_plain(ops, case):
    """GEGLU on split-K slabs of the packed (value | gate) projection (attention.py:37-50), and the plain reduce + epilogue."""
    from dsml_thesis_amd import lib as L
    M, N, nslab = case
    if nslab > 1:
        ws, prod = _slabs(ops, M, 256, N, nslab, geglu=True)      # columns in reference order in `prod`
    else:
        prod = rnd(40, M, N).double()
        inner = N // 2                                               # pack the plain tensor the way pack_geglu packs columns
        val, gate = prod[:, :inner].reshape(M, inner // 32, 1, 32), prod[:, inner:].reshape(M, inner // 32, 1, 32)
        ws = torch.cat([val, gate], 2).reshape(M, N).float().cuda()
    b = 0.1 * rnd(41, N)
    _, bp = ops.pack_geglu(torch.zeros(N, 8, device="cuda"), b.cuda())
    v_, g_ = (prod + b.double()).chunk(2, dim=1)
    ref = v_ * F.gelu(g_)
    out = torch.empty(M, N // 2, device="cuda")
    ops.post(ops.make_post_args(ws, M, N, M, nslab=nslab, bias=bp, geglu=True, raw_out=out))
    close(out, ref.float())
    if nslab > 1:
        return
    res = rnd(43, M, N)
    out2 = torch.empty(M, N, device="cuda")
    rc = res.cuda()
    ops.post(ops.make_post_args(prod.float().cuda(), M, N, M, residual=rc, raw_out=out2))
    close(out2, (prod + res.double()).float(), 1e-6, 1e-6)


def test_post_rejects_bad_arguments(ops):
    from dsml_thesis_amd import lib as L
    x = torch.zeros(64, 160, device="cuda")
    with pytest.raises(L.LdmkError, match="nothing to write"):
        ops.post(ops.make_post_args(x, 64, 160, 64))
    with pytest.raises(L.LdmkError, match="groups"):
        ops.post(ops.make_post_args(x, 64, 160, 64, norm=L.POST_GROUPNORM, gamma=x, beta=x, norm_out=x, groups=7))
    with pytest.raises(L.LdmkError, match="raw_slabs"):
        a = ops.make_igemm_args(64, 160, 160, x, 160, x, x, 160, 64, raw_slabs=True, splitk=1, tile_cfg=4)
        ops.igemm(a)


@pytest.mark.parametrize("case", [(1, 1024, 5, 1), (1, 256, 10, 3), (1, 64, 20, 6), (2, 4096, 5, 1), (1, 100, 3, 2), (3, 40, 2, 1)])
def test_attn_self_small(ops, case):
    """CrossAttention(context=None) softmax(Q K^T d^-1/2) V (attention.py:170-193) for small problems: keys split over the
    waves of a workgroup, merged in LDS; ragged token counts; qkv given as raw split-K slabs summed on load."""
    from dsml_thesis_amd import lib as L
    n, tokens, heads, nslab = case
    C = heads * 32
    slabs = torch.stack([rnd(60 + i, n * tokens, 3 * C) * (0.8 if i == 0 else 0.3) for i in range(nslab)])
    qkv = slabs.double().sum(0)
    q, k, v = (t.reshape(n, tokens, heads, 32).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=1))
    att = torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, dim=-1) @ v
    ref = att.permute(0, 2, 1, 3).reshape(n * tokens, C)
    sd = slabs.cuda()
    out = torch.empty(n * tokens, C, device="cuda")
    L.call("ldmk_attn_self_small", sd.data_ptr(), nslab, n * tokens * 3 * C, out.data_ptr(), n, tokens, heads, 32 ** -0.5,
           torch.cuda.current_stream().cuda_stream)
    close(out, ref.float(), 1e-4, 1e-4)
    out2 = torch.empty_like(out)
    L.call("ldmk_attn_self_small", sd.data_ptr(), nslab, n * tokens * 3 * C, out2.data_ptr(), n, tokens, heads, 32 ** -0.5,
           torch.cuda.current_stream().cuda_stream)
    assert torch.equal(out, out2)
    if nslab == 1:      # and it agrees with the staged kernel of the batched path
        out3 = torch.empty_like(out)
        L.call("ldmk_attn_self", sd.data_ptr(), out3.data_ptr(), n, tokens, heads, 32 ** -0.5, torch.cuda.current_stream().cuda_stream)
        close(out, out3, 2e-5, 2e-5)


@pytest.mark.parametrize("n,cin,cout,h,w,norm", [(1, 160, 3, 32, 32, True), (1, 160, 4, 64, 64, True), (2, 128, 3, 17, 5, True),
                                                 (1, 36, 1, 7, 16, False), (1, 160, 2, 1, 1, True), (1, 320, 4, 8, 8, False)])
def test_conv3x3_out_small(ops, n, cin, cout, h, w, norm):
    """openaimodel.py:683-685 (`out`: GroupNorm32 + SiLU + conv3x3 to the latent channels) with 4x4-pixel workgroups and 16
    lanes per pixel: ragged images, channel counts that do not fill the 16 lanes evenly, with and without the norm."""
    x = rnd(170, n, cin, h, w) * 1.4
    wt, b = rnd(171, cout, cin, 3, 3) / np.sqrt(9 * cin), rnd(172, cout)
    xs = nhwc(x)
    coef, ref_in = None, x
    if norm:
        gamma, beta = 1 + 0.1 * rnd(173, cin), 0.1 * rnd(174, cin)
        coef = ops.gn_coef(xs, None, n, h * w, gamma.cuda(), beta.cuda(), 1e-5)
        ref_in = F.silu(F.group_norm(x, 32, gamma, beta, 1e-5))
    wp, bd = ops.pack_conv3x3_narrow(wt.cuda()), b.cuda()
    y = ops.conv3x3_out(xs, coef, wp, bd, cout, small=True)
    close(y, F.conv2d(ref_in.double(), wt.double(), b.double(), padding=1).float(), 1e-4, 1e-4)
    close(y, ops.conv3x3_out(xs, coef, wp, bd, cout), 2e-5, 2e-5)


@pytest.mark.parametrize("cfg,splitk", [(13, 1), (15, 4), (18, 2), (14, 3), (19, 1)])
@pytest.mark.parametrize("case", [(1, 320, 160, 640, 8, 8), (2, 160, 160, 320, 16, 16), (1, 640, 0, 320, 5, 7)])
def test_slab_gemm_conv_with_fused_skip_connection(ops, cfg, splitk, case):
    """ResBlock tail (openaimodel.py:241,275): conv3x3(h) + skip_connection(cat(x, skip)) as ONE GEMM -- the 1x1 projection
    of the (two-source) block input rides along as extra K columns of the 3x3 convolution."""
    from dsml_thesis_amd import lib as L
    n, c0, c1, cout, h, w = case
    cin = c0 + c1
    hmid, x = rnd(10, n, cout, h, w), rnd(11, n, cin, h, w)
    wc, ws_ = rnd(12, cout, cout, 3, 3) / np.sqrt(9 * cout), rnd(13, cout, cin, 1, 1) / np.sqrt(cin)
    b = 0.1 * rnd(14, cout)
    ref = F.conv2d(hmid.double(), wc.double(), None, padding=1) + F.conv2d(x.double(), ws_.double(), b.double())
    wp = torch.cat([ops.pack_conv3x3(wc.cuda()), ops.pack_linear(ws_.cuda())], 0).contiguous()
    wf = ops.pack_wfrag(wp)
    xs = nhwc(x)
    x0, x1 = (xs[..., :c0].contiguous(), xs[..., c0:].contiguous()) if c1 else (xs, None)
    hd, bd = nhwc(hmid), b.cuda()
    M, K = n * h * w, 9 * cout + cin
    out = torch.empty(n, h, w, cout, device="cuda")
    a = ops.make_igemm_args(M, cout, 9 * cout, hd, cout, wp, out, cout, h * w, conv=(h, w, h, w, 1, 1, 0), bias=bd, w_frag=wf)
    a.K = K
    a.skip_a0, a.skip_c0 = x0.data_ptr(), c0
    a.skip_a1, a.skip_c1 = (0 if x1 is None else x1.data_ptr()), c1
    tm, tn, nw = SLAB_TILES[cfg]
    if cout % (32 * tn) or nw * splitk > K // 8:
        pytest.skip("tile does not fit this shape")
    _run(ops, a, cfg, splitk, M, cout)
    close(nchw(out), ref.float())
    a.tile_cfg = 4
    with pytest.raises(L.LdmkError, match="fused skip"):
        ops.igemm(a)
