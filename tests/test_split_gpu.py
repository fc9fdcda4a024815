"""GPU: LDMK_COMPUTE_BF16X3 -- fp32-accurate GEMMs on the bf16 matrix cores (include/ldmk.h).  Every fp32 operand is the exact
sum of three bf16 values; six of the nine partial products are accumulated in fp32.  The bar, stated: the result is as close
to the float64 product as the fp32 matrix-core form (LDMK_COMPUTE_F32) is -- the same accuracy class, not a reduced one."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rnd
from test_ops_gpu import close, nchw, nhwc, ops  # noqa: F401  (the `ops` fixture)

pytestmark = pytest.mark.gpu


def _err(y, ref):
    """(max, rms) error against the float64 reference"""
    d = y.detach().cpu().double() - ref
    return d.abs().max().item(), d.pow(2).mean().sqrt().item()


def _same_class(e32, e3, floor):
    """The split arithmetic is in the accuracy class of the fp32 matrix-core form: its RMS error is within 1.5x (measured:
    0.6-1.1x, tools/x3_probe.py) and its worst element -- a tail statistic of ~1e5 samples -- within 3x."""
    assert e3[1] <= max(1.5 * e32[1], 0.3 * floor), (e32, e3)
    assert e3[0] <= max(3.0 * e32[0], floor), (e32, e3)


def test_pack_wsplit_is_an_exact_split(ops):
    g = torch.Generator().manual_seed(5)
    w = torch.randn(100, 72, generator=g) * torch.exp2(torch.randint(-20, 20, (100, 72), generator=g).float())
    w[0, :8] = torch.tensor([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 3.0e38, 1.17549435e-38 * 4096, 255.99998, -0.1])
    wc = w.cuda().contiguous()
    s = ops.pack_wsplit(wc)                                  # [1][3][N][ld]
    assert s.shape == (1, 3, 72, 104) and s.dtype == torch.bfloat16
    parts = s[0].double().cpu()                              # hi, mid, lo
    back = (parts[0] + parts[1] + parts[2])[:, :100].t()
    assert torch.equal(back, w.double()), "hi + mid + lo must reproduce every fp32 weight exactly"
    assert torch.count_nonzero(s[0][:, :, 100:]).item() == 0


@pytest.mark.parametrize("cfg", [1, 2, 4, 5])
@pytest.mark.parametrize("M,K,N,sk", [(300, 320, 160, 1), (256, 640, 1920, 1), (4096, 160, 480, 1), (1024, 2560, 640, 3),
                                      (64, 1280, 1280, 4)])
def test_split_linear_matches_float64_like_the_fp32_form(ops, M, K, N, sk, cfg):
    from dsml_thesis_amd import lib as L
    x, w, b = rnd(500, M, K), rnd(501, N, K) / np.sqrt(K), 0.1 * rnd(502, N)
    res = rnd(503, M, N)
    wp = ops.pack_linear(w.cuda())
    ops.pack_wsplit(wp)
    ws = torch.empty(8 * M * N, device="cuda")
    ref = x.double() @ w.double().t() + b.double() + res.double()
    ys = []
    for compute in (L.COMPUTE_F32, L.COMPUTE_BF16X3):
        out = torch.empty(M, N, device="cuda")
        a = ops.make_igemm_args(M, N, K, x.cuda(), K, wp, out, N, M, bias=b.cuda(), residual=res.cuda(), tile_cfg=cfg, splitk=sk,
                                splitk_ws=ws, compute=compute)
        ops.igemm(a)
        ys.append(out)
    e32, e3 = _err(ys[0], ref), _err(ys[1], ref)
    _same_class(e32, e3, 2e-6)
    close(ys[1], ref.float(), 3e-6, 3e-6)


@pytest.mark.parametrize("case", [(2, 160, 320, 16, 16, 1), (2, 64, 96, 9, 7, 1), (1, 160, 160, 16, 16, 2), (3, 640, 640, 8, 8, 1),
                                  (2, 320, 160, 32, 32, 1)])
def test_split_conv3x3_with_groupnorm_silu_prologue(ops, case):
    from dsml_thesis_amd import lib as L
    n, cin, cout, h, w, stride = case
    x, wt, b = rnd(510, n, cin, h, w), rnd(511, cout, cin, 3, 3) / np.sqrt(9 * cin), 0.1 * rnd(512, cout)
    scale, shift = 1.0 + 0.2 * rnd(513, n, cin), 0.3 * rnd(514, n, cin)
    coef = torch.stack([scale, shift], 1).contiguous().cuda()
    wp = ops.pack_conv3x3(wt.cuda())
    ops.pack_wsplit(wp)
    xa = F.silu(x.double() * scale.double()[:, :, None, None] + shift.double()[:, :, None, None])
    ref = F.conv2d(xa, wt.double(), b.double(), stride=stride, padding=1)
    y32 = ops.conv3x3(nhwc(x), wp, b.cuda(), stride=stride, coef=coef)
    y3 = ops.conv3x3(nhwc(x), wp, b.cuda(), stride=stride, coef=coef, compute=L.COMPUTE_BF16X3)
    e32, e3 = _err(nchw(y32), ref), _err(nchw(y3), ref)
    _same_class(e32, e3, 3e-6)


def test_split_geglu_with_folded_layernorm(ops):
    from dsml_thesis_amd import lib as L
    M, K, inner = 512, 320, 1280
    x = rnd(520, M, K) + 0.5
    w, b = rnd(521, 2 * inner, K) / np.sqrt(K), 0.1 * rnd(522, 2 * inner)
    gamma, beta = 1.0 + 0.1 * rnd(523, K), 0.1 * rnd(524, K)
    wp, bp = ops.pack_geglu(w.cuda(), b.cuda())
    w2, cs, b2 = ops.fold_layernorm(wp, gamma.cuda(), beta.cuda(), bp)
    ops.pack_wsplit(w2)
    st = ops.ln_stats(x.cuda())
    xn = F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5)
    hcat = xn @ w.double().t() + b.double()
    ref = hcat[:, :inner] * F.gelu(hcat[:, inner:])
    y32 = ops.linear(x.cuda(), w2, b2, row_stats=st, ln_colsum=cs, geglu=True)
    y3 = ops.linear(x.cuda(), w2, b2, row_stats=st, ln_colsum=cs, geglu=True, compute=L.COMPUTE_BF16X3)
    e32, e3 = _err(y32, ref), _err(y3, ref)
    _same_class(e32, e3, 3e-6)


def test_split_batched_gemm(ops):
    """the Winograd form: 16 independent [M][K] x [K][N] products in one launch."""
    from dsml_thesis_amd import lib as L
    B, M, K, N = 16, 256, 320, 640
    a = rnd(530, B, M, K)
    w = rnd(531, B, K, N) / np.sqrt(K)
    ac, wc = a.cuda().contiguous(), w.cuda().contiguous()
    ops.pack_wsplit(wc, batch=B)
    ref = torch.bmm(a.double(), w.double())
    ws = torch.empty(4 * B * M * N, device="cuda")
    outs = []
    for compute in (L.COMPUTE_F32, L.COMPUTE_BF16X3):
        out = torch.empty(B, M, N, device="cuda")
        ar = ops.make_igemm_args(M, N, K, ac, K, wc, out, N, M, batch=B, a_bstride=M * K, w_bstride=K * N, out_bstride=M * N,
                                 tile_cfg=5, splitk=2, splitk_ws=ws, compute=compute)
        ops.igemm(ar)
        outs.append(out)
    e32, e3 = _err(outs[0], ref), _err(outs[1], ref)
    _same_class(e32, e3, 2e-6)


def test_split_rejects_what_it_cannot_run(ops):
    from dsml_thesis_amd import lib as L
    x, w = rnd(540, 64, 64).cuda(), rnd(541, 64, 64).cuda().contiguous()
    out = torch.empty(64, 64, device="cuda")
    with pytest.raises(ValueError):
        ops.make_igemm_args(64, 64, 64, x, 64, w, out, 64, 64, compute=L.COMPUTE_BF16X3)        # no split images registered
    ops.pack_wsplit(w)
    a = ops.make_igemm_args(64, 64, 64, x, 64, w, out, 64, 64, compute=L.COMPUTE_BF16X3, tile_cfg=9)
    assert L.load().ldmk_igemm_check(__import__("ctypes").byref(a)) != 0                         # row-GEMM tiles are fp32 only


@pytest.mark.parametrize("n,tokens,heads", [(2, 1024, 5), (1, 4096, 5), (3, 256, 10), (2, 64, 20), (2, 100, 2), (1, 130, 1)])
def test_split_attention_matches_float64_like_the_fp32_kernel(ops, n, tokens, heads):
    C_ = heads * 32
    qkv = (1.5 * rnd(550, n * tokens, 3 * C_)).cuda()
    q, k, v = [t.reshape(n, tokens, heads, 32).permute(0, 2, 1, 3).double() for t in qkv.cpu().split(C_, dim=1)]
    att = torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1) @ v
    ref = att.permute(0, 2, 1, 3).reshape(n * tokens, C_)
    y32 = ops.attn_self(qkv, n, tokens, heads)
    y3 = ops.attn_self(qkv, n, tokens, heads, x3=True)
    e32, e3 = _err(y32, ref), _err(y3, ref)
    _same_class(e32, e3, 2e-6)
    close(y3, ref.float(), 3e-6, 3e-6)


# ---- the warp-specialised tiles (csrc/igemm_ws.hip, tile_cfg 21 = 256x160, 22 = 256x128): producer / consumer waves, two
# 16-deep LDS stages.  Every accumulator sees the same sequence of matrix instructions as in igemm_kernel<BF = 3> with a
# 32-deep slice per iteration (tile_cfg 5 / 1), so the results are BITWISE equal to those tiles at the same split-K.
@pytest.mark.parametrize("M,K,N,sk,ws_cfg,ref_cfg", [(300, 320, 160, 1, 21, 5), (4096, 160, 480, 1, 21, 5), (1024, 2560, 640, 4, 21, 5),
                                                     (520, 640, 1920, 1, 22, 1), (64, 1280, 1280, 5, 22, 1), (8192, 160, 160, 1, 21, 5),
                                                     (130, 96, 224, 1, 21, 5), (130, 96, 100, 3, 22, 1)])
def test_warp_specialised_tiles_are_bitwise_the_lds_tiled_split_form(ops, M, K, N, sk, ws_cfg, ref_cfg):
    from dsml_thesis_amd import lib as L
    x, w, b = rnd(600, M, K), rnd(601, N, K) / np.sqrt(K), 0.1 * rnd(602, N)
    res = rnd(603, M, N)
    wp = ops.pack_linear(w.cuda())
    ops.pack_wsplit(wp)
    ws = torch.empty(8 * M * N, device="cuda")
    rows = 64 if M % 64 == 0 else M
    vec = rnd(604, -(-M // rows), N).cuda()
    outs = []
    for cfg in (ref_cfg, ws_cfg):
        out = torch.full((M, N), float("nan"), device="cuda")
        xc, bc, rc = x.cuda(), b.cuda(), res.cuda()       # (named: the args hold raw pointers, and `st` below is allocated before the launch)
        a = ops.make_igemm_args(M, N, K, xc, K, wp, out, N, rows, bias=bc, residual=rc, batch_vec=vec,
                                batch_vec_ld=N, tile_cfg=cfg, splitk=sk, splitk_ws=ws, compute=L.COMPUTE_BF16X3)
        if M % 32 == 0 and rows % 32 == 0:
            st = torch.zeros(M // 32, N, 3, device="cuda")
            a.stats_out = st.data_ptr()
            out._st = st
        ops.igemm(a)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    if hasattr(outs[0], "_st"):
        assert torch.equal(outs[0]._st, outs[1]._st)
    ref = x.double() @ w.double().t() + b.double() + res.double() + vec.cpu().double().repeat_interleave(rows, 0)[:M]
    close(outs[1], ref.float(), 6e-6, 6e-6)          # (four fp32 terms of magnitude ~1-4 summed in the epilogue)


@pytest.mark.parametrize("case", [(2, 160, 320, 16, 16, 1), (2, 64, 96, 9, 7, 1), (1, 160, 160, 16, 16, 2), (3, 640, 640, 8, 8, 1),
                                  (2, 320, 160, 32, 32, 1)])
def test_warp_specialised_conv3x3_with_prologue_and_two_sources(ops, case):
    from dsml_thesis_amd import lib as L
    n, cin, cout, h, w, stride = case
    x, wt, b = rnd(610, n, cin, h, w), rnd(611, cout, cin, 3, 3) / np.sqrt(9 * cin), 0.1 * rnd(612, cout)
    scale, shift = 1.0 + 0.2 * rnd(613, n, cin), 0.3 * rnd(614, n, cin)
    coef = torch.stack([scale, shift], 1).contiguous().cuda()
    wp = ops.pack_conv3x3(wt.cuda())
    ops.pack_wsplit(wp)
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    xs = nhwc(x)
    two = cin % 64 == 0                                     # also as a channel concat of two tensors
    outs = []
    for cfg in (5, 21):
        out = torch.full((n, oh, ow, cout), float("nan"), device="cuda")
        if two:
            x0, x1 = xs[..., :cin // 2].contiguous(), xs[..., cin // 2:].contiguous()
            a = ops.make_igemm_args(n * oh * ow, cout, 9 * cin, x0, cin // 2, wp, out, cout, oh * ow, a1=x1, c1=cin // 2,
                                    conv=(h, w, oh, ow, stride, 1, 0), tf=L.TF_AFFINE_SILU, tf_coef=coef, bias=b.cuda(), tile_cfg=cfg,
                                    splitk=1, compute=L.COMPUTE_BF16X3)
        else:
            a = ops.make_igemm_args(n * oh * ow, cout, 9 * cin, xs, cin, wp, out, cout, oh * ow, conv=(h, w, oh, ow, stride, 1, 0),
                                    tf=L.TF_AFFINE_SILU, tf_coef=coef, bias=b.cuda(), tile_cfg=cfg, splitk=1, compute=L.COMPUTE_BF16X3)
        ops.igemm(a)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    xa = F.silu(x.double() * scale.double()[:, :, None, None] + shift.double()[:, :, None, None])
    ref = F.conv2d(xa, wt.double(), b.double(), stride=stride, padding=1)
    close(nchw(outs[1]), ref.float(), 5e-6, 5e-6)


def test_warp_specialised_geglu_folded_layernorm_and_batched_planes(ops):
    from dsml_thesis_amd import lib as L
    M, K, inner = 512, 320, 1280
    x = rnd(620, M, K) + 0.5
    w, b = rnd(621, 2 * inner, K) / np.sqrt(K), 0.1 * rnd(622, 2 * inner)
    gamma, beta = 1.0 + 0.1 * rnd(623, K), 0.1 * rnd(624, K)
    wp, bp = ops.pack_geglu(w.cuda(), b.cuda())
    w2, cs, b2 = ops.fold_layernorm(wp, gamma.cuda(), beta.cuda(), bp)
    ops.pack_wsplit(w2)
    st = ops.ln_stats(x.cuda())
    y1 = ops.linear(x.cuda(), w2, b2, row_stats=st, ln_colsum=cs, geglu=True, compute=L.COMPUTE_BF16X3, tile_cfg=1)
    y22 = ops.linear(x.cuda(), w2, b2, row_stats=st, ln_colsum=cs, geglu=True, compute=L.COMPUTE_BF16X3, tile_cfg=22)
    assert torch.equal(y1, y22)
    # unfolded LayerNorm prologue (the training form) through the producer waves
    wl = ops.pack_linear((rnd(625, 160, K) / np.sqrt(K)).cuda())
    ops.pack_wsplit(wl)
    y5 = ops.linear(x.cuda(), wl, None, row_stats=st, ln_gamma=gamma.cuda(), ln_beta=beta.cuda(), compute=L.COMPUTE_BF16X3, tile_cfg=5)
    y21 = ops.linear(x.cuda(), wl, None, row_stats=st, ln_gamma=gamma.cuda(), ln_beta=beta.cuda(), compute=L.COMPUTE_BF16X3, tile_cfg=21)
    assert torch.equal(y5, y21)
    # 16 independent products in one launch (the Winograd form), split-K 2
    B, Mb, Kb, Nb = 16, 256, 320, 640
    a_, w_ = rnd(630, B, Mb, Kb).cuda().contiguous(), (rnd(631, B, Kb, Nb) / np.sqrt(Kb)).cuda().contiguous()
    ops.pack_wsplit(w_, batch=B)
    wsb = torch.empty(4 * B * Mb * Nb, device="cuda")
    outs = []
    for cfg in (5, 21):
        out = torch.empty(B, Mb, Nb, device="cuda")
        ar = ops.make_igemm_args(Mb, Nb, Kb, a_, Kb, w_, out, Nb, Mb, batch=B, a_bstride=Mb * Kb, w_bstride=Kb * Nb, out_bstride=Mb * Nb,
                                 tile_cfg=cfg, splitk=2, splitk_ws=wsb, compute=L.COMPUTE_BF16X3)
        ops.igemm(ar)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    # and what they cannot run is refused with a message
    bad = ops.make_igemm_args(M, 2 * inner, K, x.cuda(), K, w2, torch.empty(M, inner, device="cuda"), inner, M, epi=L.EPI_GEGLU,
                              compute=L.COMPUTE_BF16X3, tile_cfg=21)
    assert L.load().ldmk_igemm_check(__import__("ctypes").byref(bad)) != 0 and b"GEGLU" in L.load().ldmk_last_error()


@pytest.mark.parametrize("cfg", [1, 2, 4, 5])
@pytest.mark.parametrize("M,K,N,geglu", [(512, 320, 960, False), (300, 160, 480, False), (4096, 640, 2560, True), (1024, 160, 1280, True)])
def test_pre_split_activations_are_bitwise_the_in_kernel_split(ops, M, K, N, geglu, cfg):
    """ldmk_ln_stats_split writes the rows as three bf16 images next to their statistics; the bf16x3 GEMM then copies its A
    operand (a_split) instead of splitting it once per N-tile -- same split, same products, same bits."""
    from dsml_thesis_amd import lib as L
    if geglu and cfg not in (1, 2):
        pytest.skip("GEGLU needs an even-TN tile")
    x = rnd(700, M, K) + 0.3
    gamma, beta = 1.0 + 0.1 * rnd(701, K), 0.1 * rnd(702, K)
    if geglu:
        w, b = rnd(703, N, K) / np.sqrt(K), 0.1 * rnd(704, N)
        wp, bp = ops.pack_geglu(w.cuda(), b.cuda())
    else:
        w, b = rnd(703, N, K) / np.sqrt(K), 0.1 * rnd(704, N)
        wp, bp = ops.pack_linear(w.cuda()), b.cuda()
    w2, cs, b2 = ops.fold_layernorm(wp, gamma.cuda(), beta.cuda(), bp)
    ops.pack_wsplit(w2)
    xc = x.cuda()
    st0 = ops.ln_stats(xc)
    xs = torch.zeros(3, M, K, device="cuda", dtype=torch.bfloat16)
    st1 = ops.ln_stats(xc, split=xs)
    assert torch.equal(st0, st1)
    assert torch.equal(xs.double().sum(0).cpu(), x.double())                     # exact split
    y0 = ops.linear(xc, w2, b2, row_stats=st0, ln_colsum=cs, geglu=geglu, compute=L.COMPUTE_BF16X3, tile_cfg=cfg)
    y1 = ops.linear(xc, w2, b2, row_stats=st0, ln_colsum=cs, geglu=geglu, compute=L.COMPUTE_BF16X3, tile_cfg=cfg, a_split=xs)
    assert torch.equal(y0, y1)
    # refused where it cannot apply: f32 arithmetic, the warp-specialised tiles, a staging prologue
    import ctypes
    for kw in (dict(compute=L.COMPUTE_F32), dict(compute=L.COMPUTE_BF16X3, tile_cfg=21)):
        a = ops.make_igemm_args(M, N, K, xc, K, w2, y0, y0.shape[1], M, a_split=xs, **kw)
        assert L.load().ldmk_igemm_check(ctypes.byref(a)) != 0


@pytest.mark.parametrize("n,tokens,heads", [(2, 4096, 5), (1, 1024, 10), (3, 256, 20), (1, 960, 5), (2, 60, 5), (1, 4, 5), (1, 129, 10),
                                            (1, 2100, 5), (1, 2081, 3), (1, 2304, 2), (1, 2049, 1)])
def test_presplit_attention_is_bitwise_the_split_attention(ops, n, tokens, heads):
    """ldmk_attn_self_x3p: K / V split once by a pre-pass into MFMA-operand-order planes, tiles moved to LDS by LDS-DMA.  Same
    split values and the same instruction sequence per accumulator as ldmk_attn_self_x3: equal bit for bit, ragged token counts
    (partial 64-key tiles, partial 128-query workgroups) included.  From 2048 tokens a wave owns two 32-query blocks
    (attn_x3p_fwd_kernel<2>): a partial second block, a second block of one query, an absent second block."""
    qkv = (rnd(560, n * tokens, 3 * heads * 32) * 1.5).cuda()
    ref = ops.attn_self(qkv, n, tokens, heads, x3=True)
    out = ops.attn_self(qkv, n, tokens, heads, presplit=True)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("n,tokens,heads", [(2, 1024, 5), (1, 256, 20), (1, 4096, 5)])
def test_presplit_attention_writes_its_result_in_the_ps_layout(ops, n, tokens, heads):
    """ldmk_attn_self_x3p_ps: the attention result written straight from the accumulators in the PS layout (the pre-split A
    operand of attn1.to_out) -- exactly pack_ps of the fp32 result, with or without the fp32 copy."""
    from dsml_thesis_amd import lib as L
    C_ = heads * 32
    qkv = (rnd(570, n * tokens, 3 * C_) * 1.2).cuda()
    ref = ops.attn_self(qkv, n, tokens, heads, presplit=True)
    kv = torch.empty(L.load().ldmk_attn_kv_split_bytes(n, tokens, heads), device="cuda", dtype=torch.uint8)
    for with_fp32 in (True, False):
        out = torch.zeros(n * tokens, C_, device="cuda")
        ps = ops.ps_empty(n * tokens, C_)
        L.call("ldmk_attn_self_x3p_ps", qkv.data_ptr(), kv.data_ptr(), out.data_ptr() if with_fp32 else 0, ps.data_ptr(), n, tokens, heads,
               32 ** -0.5, ops.stream())
        assert torch.equal(ps, ops.pack_ps(ref))
        assert torch.equal(out, ref) if with_fp32 else out.abs().max().item() == 0.0
    with pytest.raises(L.LdmkError, match="tokens"):
        L.call("ldmk_attn_self_x3p_ps", qkv.data_ptr(), kv.data_ptr(), 0, ps.data_ptr(), 1, 60, heads, 0.1, ops.stream())
