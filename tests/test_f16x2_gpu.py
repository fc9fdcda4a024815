"""GPU: the F16X2 arithmetic (include/ldmk.h) -- fp32-accurate products from THREE fp16 matrix instructions per term.  An operand,
scaled by a power of two into fp16's range, is hi + lo with hi = fp16(x'), lo = fp16(x' - hi): 22 significand bits, the residual
(<= 2^-23 |x'|, one fp32 ulp) is of the size of an fp32 rounding error; hi hi + hi lo + lo hi accumulate in fp32.  The bar, stated: against
float64 the result is in the accuracy class of the fp32 matrix-core form -- RMS error within 2x (measured and simulated:
1.0-1.7x, the larger figure at the shortest K), worst element within 3x -- and an operand outside the scaled fp16 range raises
the range flag instead of producing infinities silently."""
import numpy as np
import pytest
import torch

from conftest import rnd
from test_ops_gpu import close, ops  # noqa: F401  (the `ops` fixture)
from test_split_gpu import _err

pytestmark = pytest.mark.gpu


def _f16x2_class(e32, eh, floor):
    assert eh[1] <= max(2.0 * e32[1], 0.3 * floor), (e32, eh)
    assert eh[0] <= max(3.0 * e32[0], floor), (e32, eh)


def _attention_ref(qkv, n, tokens, heads):
    C_ = heads * 32
    q, k, v = [t.reshape(n, tokens, heads, 32).permute(0, 2, 1, 3).double() for t in qkv.cpu().split(C_, dim=1)]
    att = torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1) @ v
    return att.permute(0, 2, 1, 3).reshape(n * tokens, C_)


@pytest.mark.parametrize("n,tokens,heads", [(2, 1024, 5), (1, 4096, 5), (3, 256, 10), (2, 64, 20), (2, 100, 2), (1, 130, 1),
                                            (1, 2100, 5), (1, 2081, 3), (1, 2049, 1)])
def test_f16x2_attention_matches_float64_like_the_fp32_kernel(ops, n, tokens, heads):
    """ldmk_attn_self_h2 against float64 next to the f32 matrix-core kernel; ragged token counts (partial key tiles, partial
    query blocks, the two-blocks-per-wave form from 2048 tokens)."""
    qkv = (1.5 * rnd(550, n * tokens, 3 * heads * 32)).cuda()
    ref = _attention_ref(qkv, n, tokens, heads)
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    y32 = ops.attn_self(qkv, n, tokens, heads)
    yh = ops.attn_self(qkv, n, tokens, heads, h2_flag=flag)
    assert int(flag.item()) == 0
    _f16x2_class(_err(y32, ref), _err(yh, ref), 2e-6)
    close(yh, ref.float(), 3e-6, 3e-6)


def test_f16x2_attention_peaked_and_flat_rows(ops):
    """Rows whose softmax is one-hot (large score gaps) and rows that are flat (tiny scores): the scaled-domain bookkeeping
    (scores x 2^12, probabilities x 2^14) holds at both ends."""
    n, tokens, heads = 1, 512, 2
    for mult in (1e-3, 12.0):
        qkv = (mult * rnd(551, n * tokens, 3 * heads * 32)).cuda()
        qkv[:, 2 * heads * 32:] = rnd(552, n * tokens, heads * 32).cuda()       # V of ordinary size
        ref = _attention_ref(qkv, n, tokens, heads)
        flag = torch.zeros(1, device="cuda", dtype=torch.int32)
        y32 = ops.attn_self(qkv, n, tokens, heads)
        yh = ops.attn_self(qkv, n, tokens, heads, h2_flag=flag)
        assert int(flag.item()) == 0
        _f16x2_class(_err(y32, ref), _err(yh, ref), 2e-6)


@pytest.mark.parametrize("order", ["rising", "falling", "steps", "peaked", "overflow"])
@pytest.mark.parametrize("tokens", [2048, 4096])
def test_f16x2_attention_lazy_running_maximum(ops, tokens, order):
    """The pipelined key loop (tokens % 256 == 0, two query blocks per wave) takes each 32-key block's probabilities against the
    maximum the row already has and raises it only when the block's row sum says so (csrc/attention_bf16.hip, h2p_step_lazy).
    Scores that RISE with the key index cross that bound over and over -- by a little per block, by whole binary orders at a step
    -- and scores that fall never do; one-hot rows (4 x random: score gaps of tens of binary orders) leave every other probability
    at the bottom of fp16's range.  Same bar as the exact-maximum kernel: the accuracy class of the fp32 kernel against float64.
    "overflow" (12 x random: a key 120 binary orders above everything the row has seen overflows the fp32 exponential): outside the
    kernel's range -- the range flag goes up and the result stays finite, like an operand outside fp16's range."""
    n, heads = 1, 2
    C_ = heads * 32
    g = torch.Generator().manual_seed(560)
    q = torch.randn(n * tokens, C_, generator=g)
    v = torch.randn(n * tokens, C_, generator=g)
    t = torch.arange(tokens, dtype=torch.float32)
    if order in ("peaked", "overflow"):
        mult = 4.0 if order == "peaked" else 12.0
        k = mult * torch.randn(n * tokens, C_, generator=g)
        q = mult * q
    else:
        ramp = {"rising": t / tokens, "falling": 1.0 - t / tokens, "steps": torch.floor(t / 300.0) / (tokens / 300.0)}[order]
        # every query's score against key t grows by up to ~60 (natural-log units) over the sequence: k_t = ramp_t * 60 * sqrt(32) q_dir
        qdir = torch.nn.functional.normalize(q.reshape(tokens, heads, 32).mean(0), dim=-1)          # (heads, 32)
        q = 0.3 * q + qdir.reshape(1, C_)                                                            # all queries lean the same way
        k = 0.3 * torch.randn(n * tokens, C_, generator=g) + (ramp[:, None] * 60.0 * 32 ** 0.5) * qdir.reshape(1, C_)
    qkv = torch.cat([q, k, v], dim=1).cuda()
    ref = _attention_ref(qkv, n, tokens, heads)
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    y32 = ops.attn_self(qkv, n, tokens, heads)
    yh = ops.attn_self(qkv, n, tokens, heads, h2_flag=flag)
    assert bool(torch.isfinite(yh).all())
    if order == "overflow":
        assert int(flag.item()) == 1
        return
    assert int(flag.item()) == 0
    _f16x2_class(_err(y32, ref), _err(yh, ref), 2e-6)


@pytest.mark.parametrize("where", ["k", "v", "q"])
def test_f16x2_attention_raises_the_range_flag(ops, where):
    """|K|, |V| or the pre-scaled |Q| of 1000 or more (x 2^6 leaves fp16): the flag goes up (the caller repeats the product in
    the bf16x3 arithmetic); operands just inside the range leave it down and the result stays in class."""
    n, tokens, heads = 1, 256, 2
    C_ = heads * 32
    qkv = rnd(553, n * tokens, 3 * C_).cuda()
    col = {"q": 3, "k": C_ + 5, "v": 2 * C_ + 7}[where]
    big = {"q": 1001.0 / (32 ** -0.5 * 1.4426950408889634), "k": 1001.0, "v": -1001.0}[where]
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    ops.attn_self(qkv, n, tokens, heads, h2_flag=flag)
    assert int(flag.item()) == 0
    q2 = qkv.clone()
    q2[17, col] = big
    ops.attn_self(q2, n, tokens, heads, h2_flag=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    q3 = qkv.clone()
    q3[17, col] = big * 0.99
    if where == "v":                       # a large V inside the range: the result is still fp32-class
        yh = ops.attn_self(q3, n, tokens, heads, h2_flag=flag)
        ref = _attention_ref(q3, n, tokens, heads)
        y32 = ops.attn_self(q3, n, tokens, heads)
        _f16x2_class(_err(y32, ref), _err(yh, ref), 2e-6)
    else:
        ops.attn_self(q3, n, tokens, heads, h2_flag=flag)
    assert int(flag.item()) == 0


# ---- GEMMs: LDMK_COMPUTE_F16X2 on the LDS-tiled shapes (csrc/igemm.hip, BF = 4): A scaled by 2^6 and split two ways while it is
# staged (range-checked), W pre-split by ldmk_pack_wsplit_h2 with a per-matrix scale, three fp16 MFMAs per product
import torch.nn.functional as F  # noqa: E402
from test_ops_gpu import nchw, nhwc  # noqa: E402


def _flag():
    return torch.zeros(1, device="cuda", dtype=torch.int32)


def test_pack_wsplit_h2_scales_into_range_and_splits_to_fp32_rounding(ops):
    g = torch.Generator().manual_seed(6)
    for mag in (3e-4, 0.05, 7.0, 900.0):
        w = (torch.randn(96, 72, generator=g) * mag).cuda().contiguous()
        s = ops.pack_wsplit_h2(w)                                # [1][2][N][ld]
        e = ops.split_h2_of(w.data_ptr())[5]
        assert s.shape == (1, 2, 72, 96) and s.dtype == torch.float16
        mx = w.abs().max().item() * 2.0 ** e
        assert 2.0 ** 13 <= mx < 2.0 ** 14
        back = (s[0, 0].double() + s[0, 1].double()).t().cpu() * 2.0 ** -e
        err = (back - w.double().cpu()).abs()
        # hi + lo reproduces every weight to 2^-23 relative -- one fp32 ulp (elements below 2^-3 after scaling, whose lo is
        # subnormal: an absolute 2^-25 of the scaled value)
        bound = torch.maximum(w.double().cpu().abs() * 2.0 ** -23, torch.full_like(err, 2.0 ** -25 * 2.0 ** -e))
        assert bool((err <= bound).all()), (mag, (err / bound).max().item())


@pytest.mark.parametrize("cfg", [1, 2, 4, 5])
@pytest.mark.parametrize("M,K,N,sk", [(300, 320, 160, 1), (256, 640, 1920, 1), (4096, 160, 480, 1), (1024, 2560, 640, 3),
                                      (64, 1280, 1280, 4)])
def test_f16x2_linear_matches_float64_like_the_fp32_form(ops, M, K, N, sk, cfg):
    from dsml_thesis_amd import lib as L
    x, w, b = rnd(500, M, K), rnd(501, N, K) / np.sqrt(K), 0.1 * rnd(502, N)
    res = rnd(503, M, N)
    wp = ops.pack_linear(w.cuda())
    ops.pack_wsplit_h2(wp)
    ws = torch.empty(8 * M * N, device="cuda")
    ref = x.double() @ w.double().t() + b.double() + res.double()
    flag = _flag()
    ys = []
    for compute in (L.COMPUTE_F32, L.COMPUTE_F16X2):
        out = torch.empty(M, N, device="cuda")
        xc, bc, rc = x.cuda(), b.cuda(), res.cuda()
        a = ops.make_igemm_args(M, N, K, xc, K, wp, out, N, M, bias=bc, residual=rc, tile_cfg=cfg, splitk=sk,
                                splitk_ws=ws, compute=compute, range_flag=flag)
        ops.igemm(a)
        ys.append(out)
    assert int(flag.item()) == 0
    _f16x2_class(_err(ys[0], ref), _err(ys[1], ref), 2e-6)
    close(ys[1], ref.float(), 6e-6, 6e-6)        # (the bf16x3 test holds 3e-6 here; the worst element of 655360 at K = 2560 measured 3.1e-6)


@pytest.mark.parametrize("case", [(2, 160, 320, 16, 16, 1), (2, 64, 96, 9, 7, 1), (1, 160, 160, 16, 16, 2), (3, 640, 640, 8, 8, 1),
                                  (2, 320, 160, 32, 32, 1)])
def test_f16x2_conv3x3_with_groupnorm_silu_prologue(ops, case):
    from dsml_thesis_amd import lib as L
    n, cin, cout, h, w, stride = case
    x, wt, b = rnd(510, n, cin, h, w), rnd(511, cout, cin, 3, 3) / np.sqrt(9 * cin), 0.1 * rnd(512, cout)
    scale, shift = 1.0 + 0.2 * rnd(513, n, cin), 0.3 * rnd(514, n, cin)
    coef = torch.stack([scale, shift], 1).contiguous().cuda()
    wp = ops.pack_conv3x3(wt.cuda())
    ops.pack_wsplit_h2(wp)
    xa = F.silu(x.double() * scale.double()[:, :, None, None] + shift.double()[:, :, None, None])
    ref = F.conv2d(xa, wt.double(), b.double(), stride=stride, padding=1)
    flag = _flag()
    y32 = ops.conv3x3(nhwc(x), wp, b.cuda(), stride=stride, coef=coef)
    yh = ops.conv3x3(nhwc(x), wp, b.cuda(), stride=stride, coef=coef, compute=L.COMPUTE_F16X2, range_flag=flag)
    assert int(flag.item()) == 0
    _f16x2_class(_err(nchw(y32), ref), _err(nchw(yh), ref), 3e-6)


def test_f16x2_geglu_with_folded_layernorm(ops):
    from dsml_thesis_amd import lib as L
    M, K, inner = 512, 320, 1280
    x = rnd(520, M, K) + 0.5
    w, b = rnd(521, 2 * inner, K) / np.sqrt(K), 0.1 * rnd(522, 2 * inner)
    gamma, beta = 1.0 + 0.1 * rnd(523, K), 0.1 * rnd(524, K)
    wp, bp = ops.pack_geglu(w.cuda(), b.cuda())
    w2, cs, b2 = ops.fold_layernorm(wp, gamma.cuda(), beta.cuda(), bp)
    ops.pack_wsplit_h2(w2)
    st = ops.ln_stats(x.cuda())
    xn = F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5)
    hcat = xn @ w.double().t() + b.double()
    ref = hcat[:, :inner] * F.gelu(hcat[:, inner:])
    flag = _flag()
    y32 = ops.linear(x.cuda(), w2, b2, row_stats=st, ln_colsum=cs, geglu=True)
    yh = ops.linear(x.cuda(), w2, b2, row_stats=st, ln_colsum=cs, geglu=True, compute=L.COMPUTE_F16X2, range_flag=flag)
    assert int(flag.item()) == 0
    _f16x2_class(_err(y32, ref), _err(yh, ref), 3e-6)


def test_f16x2_batched_gemm(ops):
    """the Winograd form: 16 independent [M][K] x [K][N] products in one launch (one weight scale for the batch)."""
    from dsml_thesis_amd import lib as L
    B, M, K, N = 16, 256, 320, 640
    a = rnd(530, B, M, K)
    w = rnd(531, B, K, N) / np.sqrt(K)
    ac, wc = a.cuda().contiguous(), w.cuda().contiguous()
    ops.pack_wsplit_h2(wc, batch=B)
    ref = torch.bmm(a.double(), w.double())
    ws = torch.empty(4 * B * M * N, device="cuda")
    flag = _flag()
    outs = []
    for compute in (L.COMPUTE_F32, L.COMPUTE_F16X2):
        out = torch.empty(B, M, N, device="cuda")
        ar = ops.make_igemm_args(M, N, K, ac, K, wc, out, N, M, batch=B, a_bstride=M * K, w_bstride=K * N, out_bstride=M * N,
                                 tile_cfg=5, splitk=2, splitk_ws=ws, compute=compute, range_flag=flag)
        ops.igemm(ar)
        outs.append(out)
    assert int(flag.item()) == 0
    _f16x2_class(_err(outs[0], ref), _err(outs[1], ref), 2e-6)


def test_f16x2_gemm_small_and_large_magnitudes_and_the_range_flag(ops):
    """Operand magnitudes from 0.02 to 110: the error stays in class.  Far below 2^-9 an activation's lo image is a subnormal fp16
    number and the element keeps an ABSOLUTE precision of 2^-31 (2^-25 in the scaled domain) instead of a relative one: a tensor
    that is 1e-4 everywhere comes out with an error of that floor x sqrt(K) x rms(w) -- 8 x the fp32 form's there, 3e-10 in absolute
    terms (stated in include/ldmk.h).  One element of 1000 or more raises the flag."""
    from dsml_thesis_amd import lib as L
    M, K, N = 256, 320, 160
    w = rnd(541, N, K) / np.sqrt(K)
    wp = ops.pack_linear(w.cuda())
    ops.pack_wsplit_h2(wp)
    for mag in (1e-4, 0.02, 1.0, 30.0, 500.0 / 4.5):
        x = rnd(540, M, K) * mag
        ref = x.double() @ w.double().t()
        flag = _flag()
        y32 = ops.linear(x.cuda(), wp)
        yh = ops.linear(x.cuda(), wp, compute=L.COMPUTE_F16X2, range_flag=flag)
        assert int(flag.item()) == 0, mag
        e32, eh = _err(y32, ref), _err(yh, ref)
        if mag >= 0.02:
            _f16x2_class(e32, eh, 2e-6 * mag)
        else:
            floor = 2.0 ** -31 * np.sqrt(K) * w.double().pow(2).mean().sqrt().item()
            assert eh[1] <= floor and eh[0] <= 6 * floor, (mag, e32, eh, floor)
    x = rnd(540, M, K)
    x[100, 37] = -1000.5
    flag = _flag()
    ops.linear(x.cuda(), wp, compute=L.COMPUTE_F16X2, range_flag=flag)
    assert int(flag.item()) == 1


def test_f16x2_rejects_what_it_cannot_run(ops):
    from dsml_thesis_amd import lib as L
    import ctypes
    x, w = rnd(540, 64, 64).cuda(), rnd(541, 64, 64).cuda().contiguous()
    out = torch.empty(64, 64, device="cuda")
    flag = _flag()
    with pytest.raises(ValueError):
        ops.make_igemm_args(64, 64, 64, x, 64, w, out, 64, 64, compute=L.COMPUTE_F16X2, range_flag=flag)       # no fp16 images registered
    ops.pack_wsplit_h2(w)
    with pytest.raises(ValueError):
        ops.make_igemm_args(64, 64, 64, x, 64, w, out, 64, 64, compute=L.COMPUTE_F16X2)                        # no range flag
    for cfg in (9, 21, 23):                                                                                    # row-GEMM, warp-specialised, pre-split tiles
        a = ops.make_igemm_args(64, 64, 64, x, 64, w, out, 64, 64, compute=L.COMPUTE_F16X2, range_flag=flag, tile_cfg=cfg)
        assert L.load().ldmk_igemm_check(ctypes.byref(a)) != 0
    a = ops.make_igemm_args(64, 64, 64, x, 64, w, out, 64, 64, compute=L.COMPUTE_F16X2, range_flag=flag, tile_cfg=5)
    assert L.load().ldmk_igemm_check(ctypes.byref(a)) == 0


# ---- the pre-split tiles in F16X2 (csrc/igemm_ps.hip, PL = 2): two fp16 planes per operand in 2-KiB units, LDS-DMA, three products
@pytest.mark.parametrize("rows,k", [(64, 160), (100, 64), (33, 1280)])
def test_pack_ps_h2_layout_and_split(ops, rows, k):
    x = rnd(700, rows, k) * 3.0
    flag = _flag()
    ps = ops.pack_ps(x.cuda(), h2_flag=flag)
    assert ps.numel() == -(-rows // 32) * (k // 16) * 2048 and int(flag.item()) == 0
    hi, lo = ops.unpack_ps_h2(ps, rows, k)
    xs = x.double() * 64.0
    assert torch.equal(hi.cpu(), xs.float().half().float())                              # hi = fp16(2^6 x), nearest even
    assert torch.equal(lo.cpu(), (xs.float() - hi.cpu()).half().float())                 # lo = fp16(2^6 x - hi)
    err = (hi.cpu().double() + lo.cpu().double() - xs).abs()
    assert bool((err <= torch.maximum(xs.abs() * 2.0 ** -23, torch.full_like(err, 2.0 ** -25))).all())
    x[3, 5] = 1200.0
    ops.pack_ps(x.cuda(), h2_flag=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("cfg", [23, 24, 26, 27, 31])
@pytest.mark.parametrize("M,K,N,sk", [(300, 320, 160, 1), (512, 640, 1920, 1), (4096, 160, 480, 1), (1024, 2560, 640, 3), (33, 64, 32, 1)])
def test_ps_h2_gemm_is_bitwise_the_lds_tiled_f16x2_gemm(ops, M, K, N, sk, cfg):
    """Same split values, same three products in the same order per accumulator as igemm_kernel<BF = 4>: bit for bit tile_cfg 5 at
    equal split-K (bias + per-sample vector + residual epilogue, ragged M, both epilogue forms)."""
    from dsml_thesis_amd import lib as L
    x, w, b = rnd(600, M, K), rnd(601, N, K) / np.sqrt(K), 0.1 * rnd(602, N)
    res, vec = rnd(603, M, N).cuda(), rnd(604, 3, N).cuda()
    rps = -(-M // 3)
    wp = ops.pack_linear(w.cuda())
    ops.pack_wsplit_h2(wp)
    flag = _flag()
    wps, xps = ops.pack_wps(wp, h2=True), ops.pack_ps(x.cuda(), h2_flag=flag)
    ws = torch.empty(8 * M * N, device="cuda")
    xc, bc = x.cuda(), b.cuda()
    ref = torch.empty(M, N, device="cuda")
    a = ops.make_igemm_args(M, N, K, xc, K, wp, ref, N, rps, tile_cfg=5, splitk=sk, splitk_ws=ws, compute=L.COMPUTE_F16X2, range_flag=flag,
                            bias=bc, residual=res, batch_vec=vec, batch_vec_ld=N)
    ops.igemm(a)
    out = torch.empty(M, N, device="cuda")
    a = ops.make_igemm_args(M, N, K, None, K, wp, out, N, rps, tile_cfg=cfg, splitk=sk, splitk_ws=ws, a_ps=xps, w_ps=wps, range_flag=flag,
                            bias=bc, residual=res, batch_vec=vec, batch_vec_ld=N)
    assert a.compute == L.COMPUTE_F16X2
    ops.igemm(a)
    assert int(flag.item()) == 0
    assert torch.equal(out, ref)
    if M % 32 == 0 and sk == 1:          # GroupNorm records: the lane = column form of the kernel
        rec_ref, o2 = torch.zeros(M // 32, N, 3, device="cuda"), torch.empty(M, N, device="cuda")
        a = ops.make_igemm_args(M, N, K, xc, K, wp, o2, N, M, tile_cfg=5, splitk=1, bias=bc, compute=L.COMPUTE_F16X2, range_flag=flag)
        a.stats_out = rec_ref.data_ptr()
        ops.igemm(a)
        rec, o3 = torch.zeros(M // 32, N, 3, device="cuda"), torch.empty(M, N, device="cuda")
        a = ops.make_igemm_args(M, N, K, None, K, wp, o3, N, M, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=wps, bias=bc, range_flag=flag)
        a.stats_out = rec.data_ptr()
        ops.igemm(a)
        assert torch.equal(o3, o2) and torch.equal(rec, rec_ref)


@pytest.mark.parametrize("cfg", [25, 28, 32, 33])
@pytest.mark.parametrize("M,K,N", [(512, 160, 1280), (4096, 320, 2560), (96, 640, 5120)])
def test_ps_h2_geglu_with_folded_layernorm_and_split_output(ops, M, K, N, cfg):
    """The GEGLU projection as the transformer block runs it in F16X2: the statistics pass writes the rows as fp16 planes, the GEMM
    is bitwise tile_cfg 1's F16X2 result, and out_ps is exactly pack_ps(h2) of the fp32 result (range-checked)."""
    from dsml_thesis_amd import lib as L
    x = rnd(610, M, K) + 0.5 * rnd(611, M, 1)
    w, b = rnd(612, N, K) / np.sqrt(K), 0.1 * rnd(613, N)
    g, be = 1 + 0.2 * rnd(614, K), 0.2 * rnd(615, K)
    wp, bp = ops.pack_geglu(w.cuda(), b.cuda())
    w2, cs, b2 = ops.fold_layernorm(wp, g.cuda(), be.cuda(), bp)
    ops.pack_wsplit_h2(w2)
    flag = _flag()
    xc = x.cuda()
    st, xps = ops.ln_stats_ps(xc, h2_flag=flag)
    st0 = ops.ln_stats(xc)
    close(st, st0.cpu(), 2e-6, 2e-6)                      # (another reduction order than ldmk_ln_stats)
    assert torch.equal(xps, ops.pack_ps(xc, h2_flag=flag))
    ref = torch.empty(M, N // 2, device="cuda")
    a = ops.make_igemm_args(M, N, K, xc, K, w2, ref, N // 2, M, tf=L.TF_LAYERNORM_FOLDED, row_stats=st, ln_colsum=cs, bias=b2,
                            epi=L.EPI_GEGLU, tile_cfg=1, splitk=1, compute=L.COMPUTE_F16X2, range_flag=flag)
    ops.igemm(a)
    wps = ops.pack_wps(w2, h2=True)
    for with_fp32 in (True, False):
        out = torch.zeros(M, N // 2, device="cuda")
        ops_out = ops.ps_empty(M, N // 2, h2=True)
        a = ops.make_igemm_args(M, N, K, None, K, w2, out if with_fp32 else None, N // 2, M, tf=L.TF_LAYERNORM_FOLDED, row_stats=st,
                                ln_colsum=cs, bias=b2, epi=L.EPI_GEGLU, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=wps, out_ps=ops_out, range_flag=flag)
        ops.igemm(a)
        if with_fp32:
            assert torch.equal(out, ref)
        assert torch.equal(ops_out, ops.pack_ps(ref, h2_flag=flag))
    assert int(flag.item()) == 0


def test_ps_h2_rejections(ops):
    from dsml_thesis_amd import lib as L
    import ctypes
    M, K, N = 64, 64, 64
    x, w = rnd(620, M, K).cuda(), rnd(621, K, N).cuda().contiguous()
    flag = _flag()
    xps, wps = ops.pack_ps(x, h2_flag=flag), ops.pack_wps(w, h2=True)
    out = torch.empty(M, N, device="cuda")
    for cfg in (29, 30):                   # the warp-specialised pre-split tiles exist in bf16x3 only
        a = ops.make_igemm_args(M, N, K, None, K, w, out, N, M, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=wps, range_flag=flag)
        assert L.load().ldmk_igemm_check(ctypes.byref(a)) != 0
    a = ops.make_igemm_args(M, N, K, None, K, w, out, N, M, tile_cfg=27, splitk=1, a_ps=xps, w_ps=wps, out_ps=ops.ps_empty(M, N, h2=True))
    assert L.load().ldmk_igemm_check(ctypes.byref(a)) != 0        # out_ps in F16X2 without a range flag
    a = ops.make_igemm_args(M, N, K, None, K, w, out, N, M, tile_cfg=27, splitk=1, a_ps=xps, w_ps=wps, range_flag=flag)
    assert L.load().ldmk_igemm_check(ctypes.byref(a)) == 0


@pytest.mark.parametrize("case", [(2, 320, 0, 16, 16), (1, 160, 160, 8, 8), (3, 64, 32, 4, 6), (1, 640, 0, 32, 32)])
def test_winograd_input_in_the_f16x2_ps_layout(ops, case):
    """ldmk_winograd_input_ps_h2: the 16 planes of V = B^T d B written as two fp16 planes of 2^6 V -- plane by plane exactly
    pack_ps(h2) of what ldmk_winograd_input writes (whole row blocks), range-checked."""
    from dsml_thesis_amd import lib as L
    n, c0, c1, h, w = case
    C = c0 + c1
    x0 = rnd(700, n, h, w, c0).cuda()
    x1 = rnd(701, n, h, w, c1).cuda() if c1 else None
    coef = torch.stack([1.0 + 0.2 * rnd(702, n, C), 0.3 * rnd(703, n, C)], 1).contiguous().cuda()
    tiles = n * (h // 2) * (w // 2)
    V = torch.empty(16, tiles, C, device="cuda")
    L.call("ldmk_winograd_input", x0.data_ptr(), c0, 0 if x1 is None else x1.data_ptr(), c1, coef.data_ptr(), 1, n, h, w, V.data_ptr(),
           ops.stream())
    flag = _flag()
    Vps = ops.ps_empty(tiles, C, batch=16, h2=True)
    L.call("ldmk_winograd_input_ps_h2", x0.data_ptr(), c0, 0 if x1 is None else x1.data_ptr(), c1, coef.data_ptr(), 1, n, h, w,
           Vps.data_ptr(), flag.data_ptr(), ops.stream())
    ref = ops.pack_ps(V, h2_flag=flag)
    nb = (tiles // 32) * (C // 16) * 2048
    assert torch.equal(Vps[:, :nb], ref[:, :nb]) and int(flag.item()) == 0
    x0[0, 1, 1, 3] = 5000.0
    L.call("ldmk_winograd_input_ps_h2", x0.data_ptr(), c0, 0 if x1 is None else x1.data_ptr(), c1, coef.data_ptr(), 1, n, h, w,
           Vps.data_ptr(), flag.data_ptr(), ops.stream())
    assert int(flag.item()) == 1


@pytest.mark.parametrize("case", [(2, 320, 16, 16), (1, 64, 5, 7), (1, 640, 16, 16)])
def test_upconv_gather_in_the_f16x2_ps_layout(ops, case):
    from dsml_thesis_amd import lib as L
    n, c, h, w = case
    x = rnd(710, n, h, w, c).cuda()
    pix = n * h * w
    A = torch.empty(4, pix, 4 * c, device="cuda")
    L.call("ldmk_upconv_gather", x.data_ptr(), c, n, h, w, A.data_ptr(), ops.stream())
    flag = _flag()
    Aps = ops.ps_empty(pix, 4 * c, batch=4, h2=True)
    L.call("ldmk_upconv_gather_ps_h2", x.data_ptr(), c, n, h, w, Aps.data_ptr(), flag.data_ptr(), ops.stream())
    ref = ops.pack_ps(A, h2_flag=flag)
    nb = (pix // 32) * (4 * c // 16) * 2048
    assert torch.equal(Aps[:, :nb], ref[:, :nb]) and int(flag.item()) == 0


@pytest.mark.parametrize("cfg", [23, 24, 27, 31])
def test_ps_h2_gemm_batched_planes(ops, cfg):
    """The Winograd form on the pre-split tiles in F16X2: 16 plane products in one launch, one weight scale for the batch;
    bitwise tile_cfg 5's F16X2 result."""
    from dsml_thesis_amd import lib as L
    B, M, K, N = 16, 256, 320, 640
    a_ = rnd(530, B, M, K).cuda().contiguous()
    w = (rnd(531, B, K, N) / np.sqrt(K)).cuda().contiguous()
    ops.pack_wsplit_h2(w, batch=B)
    flag = _flag()
    wps, aps = ops.pack_wps(w, batch=B, h2=True), ops.pack_ps(a_, h2_flag=flag)
    ref, out = torch.empty(B, M, N, device="cuda"), torch.empty(B, M, N, device="cuda")
    ar = ops.make_igemm_args(M, N, K, a_, K, w, ref, N, M, batch=B, a_bstride=M * K, w_bstride=K * N, out_bstride=M * N, tile_cfg=5, splitk=1,
                             compute=L.COMPUTE_F16X2, range_flag=flag)
    ops.igemm(ar)
    ar = ops.make_igemm_args(M, N, K, None, K, w, out, N, M, batch=B, w_bstride=K * N, out_bstride=M * N, tile_cfg=cfg, splitk=1,
                             a_ps=aps, w_ps=wps, range_flag=flag)
    ops.igemm(ar)
    assert torch.equal(out, ref) and int(flag.item()) == 0


@pytest.mark.parametrize("n,tokens,heads", [(2, 1024, 5), (1, 512, 20), (1, 4096, 5)])
def test_f16x2_attention_writes_its_result_in_the_f16x2_ps_layout(ops, n, tokens, heads):
    """ldmk_attn_self_h2_ps: the attention result straight from the accumulators as two fp16 planes of 2^6 x -- exactly pack_ps(h2)
    of the fp32 result, with or without the fp32 copy."""
    from dsml_thesis_amd import lib as L
    C_ = heads * 32
    qkv = (rnd(570, n * tokens, 3 * C_) * 1.2).cuda()
    flag = _flag()
    ref = ops.attn_self(qkv, n, tokens, heads, h2_flag=flag)
    kv = torch.empty(L.load().ldmk_attn_kv_split_h2_bytes(n, tokens, heads), device="cuda", dtype=torch.uint8)
    for with_fp32 in (True, False):
        out = torch.zeros(n * tokens, C_, device="cuda")
        ps = ops.ps_empty(n * tokens, C_, h2=True)
        L.call("ldmk_attn_self_h2_ps", qkv.data_ptr(), kv.data_ptr(), out.data_ptr() if with_fp32 else 0, ps.data_ptr(), flag.data_ptr(), n, tokens,
               heads, 32 ** -0.5, ops.stream())
        assert torch.equal(ps, ops.pack_ps(ref, h2_flag=flag))
        assert torch.equal(out, ref) if with_fp32 else out.abs().max().item() == 0.0
    assert int(flag.item()) == 0
    with pytest.raises(L.LdmkError, match="tokens"):
        L.call("ldmk_attn_self_h2_ps", qkv.data_ptr(), kv.data_ptr(), 0, ps.data_ptr(), flag.data_ptr(), 1, 60, heads, 0.1, ops.stream())


@pytest.mark.parametrize("cfg", [23, 27])
@pytest.mark.parametrize("n,tokens,heads", [(2, 1024, 5), (1, 4096, 5), (3, 64, 10), (1, 512, 20)])
def test_qkv_projection_writes_the_attention_kv_tiles(ops, n, tokens, heads, cfg):
    """The fused QKV projection on a pre-split F16X2 tile with attn_kv_out: the q third of the result is the fp32 result of the
    plain projection, the K / V thirds come out as the attention's pre-split tiles -- bit for bit what ldmk_attn_self_h2's pre-pass
    writes from the fp32 K / V -- and ldmk_attn_self_h2_tiles on them equals ldmk_attn_self_h2 on the fp32 projection."""
    from dsml_thesis_amd import lib as L
    C_ = heads * 32
    M, K, N = n * tokens, C_, 3 * C_
    x = rnd(800, M, K) + 0.3 * rnd(801, M, 1)
    w, b = rnd(802, N, K) / np.sqrt(K), 0.1 * rnd(803, N)
    g, be = 1 + 0.2 * rnd(804, K), 0.2 * rnd(805, K)
    wp = ops.pack_linear(w.cuda())
    w2, cs, b2 = ops.fold_layernorm(wp, g.cuda(), be.cuda(), b.cuda())
    flag = _flag()
    xc = x.cuda()
    st, xps = ops.ln_stats_ps(xc, h2_flag=flag)
    wps = ops.pack_wps(w2, h2=True)
    kw = dict(tf=L.TF_LAYERNORM_FOLDED, row_stats=st, ln_colsum=cs, bias=b2, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=wps, range_flag=flag)
    ref = torch.empty(M, N, device="cuda")
    ops.igemm(ops.make_igemm_args(M, N, K, None, K, w2, ref, N, tokens, **kw))
    out = torch.full((M, N), 7.0, device="cuda")
    kv = torch.zeros(L.load().ldmk_attn_kv_split_h2_bytes(n, tokens, heads), device="cuda", dtype=torch.uint8)
    a = ops.make_igemm_args(M, N, K, None, K, w2, out, N, tokens, attn_kv=(kv, tokens, heads), **kw)
    assert L.load().ldmk_igemm_check(__import__("ctypes").byref(a)) == 0
    ops.igemm(a)
    assert torch.equal(out[:, :C_], ref[:, :C_]) and bool((out[:, C_:] == 7.0).all())          # K / V are not stored as fp32
    att_ref = torch.empty(M, C_, device="cuda")
    kv_ref = torch.zeros_like(kv)
    L.call("ldmk_attn_self_h2", ref.data_ptr(), kv_ref.data_ptr(), att_ref.data_ptr(), flag.data_ptr(), n, tokens, heads, 32 ** -0.5, ops.stream())
    assert torch.equal(kv, kv_ref)
    att = torch.empty(M, C_, device="cuda")
    L.call("ldmk_attn_self_h2_tiles", out.data_ptr(), kv.data_ptr(), att.data_ptr(), 0, flag.data_ptr(), n, tokens, heads, 32 ** -0.5, ops.stream())
    assert torch.equal(att, att_ref) and int(flag.item()) == 0
    # refused: other tiles, split-K, ragged token counts
    for bad in (dict(tile_cfg=31), dict(splitk=2), dict(attn_kv=(kv, tokens - 32, heads))):
        kw2 = dict(kw, attn_kv=(kv, tokens, heads))
        kw2.update(bad)
        a = ops.make_igemm_args(M, N, K, None, K, w2, out, N, tokens, splitk_ws=torch.empty(4 * M * N, device="cuda") if "splitk" in bad else None, **kw2)
        assert L.load().ldmk_igemm_check(__import__("ctypes").byref(a)) != 0
