"""GPU: the pre-split bf16x3 tiles (csrc/igemm_ps.hip, tile_cfg 23..28).  Both operands reach the kernel already split into
three bf16 planes in the PS layout (include/ldmk.h) and are moved memory -> LDS by LDS-DMA loads; the kernel does no arithmetic
on them.  The bar: the SAME bits as the LDS-tiled bf16x3 form (tile_cfg 1 / 5) at equal split-K -- same products, same order --
so every accuracy statement of tests/test_split_gpu.py carries over unchanged; and the producers of the layout (ldmk_pack_ps,
ldmk_ln_stats_ps, the GEMM's own out_ps epilogue) write the exact three-way split."""
import numpy as np
import pytest
import torch

from conftest import rnd
from test_ops_gpu import close, ops  # noqa: F401  (the `ops` fixture)

pytestmark = pytest.mark.gpu

PS = {23: (256, 160), 24: (256, 320), 25: (256, 256), 26: (128, 320), 27: (128, 160), 28: (128, 256), 29: (256, 160), 30: (256, 128)}


def _split3(x):
    """the exact three-way split in torch: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest even"""
    hi = x.to(torch.bfloat16).float()
    r = x - hi
    mid = r.to(torch.bfloat16).float()
    lo = (r - mid).to(torch.bfloat16).float()
    return hi, mid, lo


@pytest.mark.parametrize("rows,k", [(64, 32), (100, 160), (4096, 640), (33, 1280)])
def test_pack_ps_is_the_exact_split_in_the_documented_layout(ops, rows, k):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(rows, k, generator=g) * torch.exp2(torch.randint(-12, 12, (rows, k), generator=g).float())
    x[0, :4] = torch.tensor([0.0, 1.0 + 2.0 ** -23, 3.0e38, -0.1])
    ps = ops.pack_ps(x.cuda())
    assert ps.numel() == ((rows + 31) // 32) * (k // 16) * 3072
    hi, mid, lo = ops.unpack_ps(ps.cpu(), rows, k)
    rh, rm, rl = _split3(x)
    assert torch.equal(hi, rh) and torch.equal(mid, rm) and torch.equal(lo, rl)
    assert torch.equal((hi.double() + mid.double() + lo.double()), x.double()), "hi + mid + lo reproduces every fp32 value"
    # weights: w[K][N] packed as X[N][K]
    w = rnd(3, k, 96).cuda()
    wh, wm, wl = ops.unpack_ps(ops.pack_wps(w).cpu(), 96, k)
    rh, rm, rl = _split3(w.t().cpu())
    assert torch.equal(wh, rh) and torch.equal(wm, rm) and torch.equal(wl, rl)


@pytest.mark.parametrize("K", [160, 320, 640, 1280])
def test_ln_stats_ps_statistics_and_planes(ops, K):
    rows = 200
    x = rnd(11, rows, K) * 1.5 + 0.3 * rnd(12, rows, 1)
    xc = x.cuda()
    st, ps = ops.ln_stats_ps(xc)
    ref_mean = x.double().mean(1)
    ref_rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5)
    assert (st[:, 0].cpu().double() - ref_mean).abs().max() < 2e-7 * max(1.0, ref_mean.abs().max().item())
    assert ((st[:, 1].cpu().double() - ref_rstd) / ref_rstd).abs().max() < 2e-6
    if K <= 1024:
        assert (st - ops.ln_stats(xc)).abs().max().item() < 1e-6       # (another summation order than ldmk_ln_stats)
    assert torch.equal(ps, ops.pack_ps(xc))
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.ln_stats_ps(xc, guard=1e9, flag=flag)
    assert flag.item() == 0
    xb = x.clone()
    xb[5] += 100.0
    ops.ln_stats_ps(xb.cuda(), guard=4.0, flag=flag)
    assert flag.item() == 1


def _ref_igemm(ops, M, N, K, x, wp, cfg, sk, ws, **kw):
    from dsml_thesis_amd import lib as L
    out = torch.empty(M, kw.pop("ncol", N), device="cuda")
    a = ops.make_igemm_args(M, N, K, x, K, wp, out, out.shape[1], kw.pop("rows_per_sample", M), tile_cfg=cfg, splitk=sk, splitk_ws=ws,
                            compute=L.COMPUTE_BF16X3, **kw)
    ops.igemm(a)
    return out


@pytest.mark.parametrize("cfg", [23, 24, 26, 27, 29, 31])
@pytest.mark.parametrize("M,K,N,sk", [(300, 320, 160, 1), (512, 640, 1920, 1), (4096, 160, 480, 1), (1024, 2560, 640, 3),
                                      (64, 1280, 320, 4), (33, 64, 32, 1)])
def test_ps_gemm_is_bitwise_the_lds_tiled_bf16x3_gemm(ops, M, K, N, sk, cfg):
    """bias + per-sample vector + residual epilogue; K split over workgroups with the reduce launch; ragged M; N not a multiple of
    the tile width; both epilogue forms of the kernel (with GroupNorm records: lane = column; without: transposed)."""
    x, w, b = rnd(600, M, K), rnd(601, N, K) / np.sqrt(K), 0.1 * rnd(602, N)
    res, vec = rnd(603, M, N).cuda(), rnd(604, 3, N).cuda()
    rps = -(-M // 3)
    wp = ops.pack_linear(w.cuda())
    ops.pack_wsplit(wp)
    wps, xps = ops.pack_wps(wp), ops.pack_ps(x.cuda())
    ws = torch.empty(8 * M * N, device="cuda")
    kw = dict(bias=b.cuda(), residual=res, batch_vec=vec, batch_vec_ld=N, rows_per_sample=rps)
    ref = _ref_igemm(ops, M, N, K, x.cuda(), wp, 5, sk, ws, **dict(kw))
    out = torch.empty(M, N, device="cuda")
    a = ops.make_igemm_args(M, N, K, None, K, wp, out, N, rps, tile_cfg=cfg, splitk=sk, splitk_ws=ws, a_ps=xps, w_ps=wps,
                            bias=b.cuda(), residual=res, batch_vec=vec, batch_vec_ld=N)
    ops.igemm(a)
    assert torch.equal(out, ref)
    close(out, (x.double() @ w.double().t() + b.double()).float() + res.cpu() + vec.cpu().repeat_interleave(rps, 0)[:M], 5e-6, 5e-6)
    if M % 32 == 0 and sk == 1:
        # GroupNorm partial records from the epilogue (the lane = column form of the kernel)
        rec_ref = torch.zeros(M // 32, N, 3, device="cuda")
        o2 = torch.empty(M, N, device="cuda")
        a = ops.make_igemm_args(M, N, K, x.cuda(), K, wp, o2, N, M, tile_cfg=5, splitk=1, bias=b.cuda(), compute=2)
        a.stats_out = rec_ref.data_ptr()
        ops.igemm(a)
        rec, o3 = torch.zeros(M // 32, N, 3, device="cuda"), torch.empty(M, N, device="cuda")
        a = ops.make_igemm_args(M, N, K, None, K, wp, o3, N, M, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=wps, bias=b.cuda())
        a.stats_out = rec.data_ptr()
        ops.igemm(a)
        assert torch.equal(o3, o2) and torch.equal(rec, rec_ref)


@pytest.mark.parametrize("cfg", [25, 28, 30, 32, 33])
@pytest.mark.parametrize("M,K,N", [(512, 160, 1280), (4096, 320, 2560), (96, 640, 5120)])
def test_ps_gemm_geglu_with_folded_layernorm_and_split_output(ops, M, K, N, cfg):
    """The GEGLU projection as the transformer block runs it: LayerNorm folded through the product (row statistics from the
    pass that also wrote the PS rows), (value, gate) column pairs, gate through GELU -- bitwise tile_cfg 1's result; and the
    result written pre-split (out_ps) is exactly pack_ps of the fp32 result, so the next GEMM reads what a separate pass would
    have produced."""
    from dsml_thesis_amd import lib as L
    x = rnd(610, M, K) + 0.5 * rnd(611, M, 1)
    w, b = rnd(612, N, K) / np.sqrt(K), 0.1 * rnd(613, N)
    g, be = 1 + 0.2 * rnd(614, K), 0.2 * rnd(615, K)
    wp, bp = ops.pack_geglu(w.cuda(), b.cuda())
    w2, cs, b2 = ops.fold_layernorm(wp, g.cuda(), be.cuda(), bp)
    ops.pack_wsplit(w2)
    xc = x.cuda()
    st, xps = ops.ln_stats_ps(xc)
    ref = torch.empty(M, N // 2, device="cuda")
    a = ops.make_igemm_args(M, N, K, xc, K, w2, ref, N // 2, M, tf=L.TF_LAYERNORM_FOLDED, row_stats=st, ln_colsum=cs, bias=b2,
                            epi=L.EPI_GEGLU, tile_cfg=1, splitk=1, compute=L.COMPUTE_BF16X3)
    ops.igemm(a)
    out = torch.empty(M, N // 2, device="cuda")
    ops_ps = ops.ps_empty(M, N // 2)
    a = ops.make_igemm_args(M, N, K, None, K, w2, out, N // 2, M, tf=L.TF_LAYERNORM_FOLDED, row_stats=st, ln_colsum=cs, bias=b2,
                            epi=L.EPI_GEGLU, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=ops.pack_wps(w2), out_ps=ops_ps)
    ops.igemm(a)
    assert torch.equal(out, ref)
    h, m, l = ops.unpack_ps(ops_ps.cpu(), M, N // 2)
    rh, rm, rl = _split3(out.cpu())
    assert torch.equal(h, rh) and torch.equal(m, rm) and torch.equal(l, rl)
    # float64 reference of the whole layer
    xn = torch.nn.functional.layer_norm(x.double(), (K,), g.double(), be.double(), 1e-5)
    y = xn @ w.double().t() + b.double()
    val, gate = y[:, :N // 2], y[:, N // 2:]
    close(out, (val * torch.nn.functional.gelu(gate)).float(), 2e-5, 2e-5)
    # PS-only output (no fp32 copy): what ff.net.2 reads
    ops_ps2 = ops.ps_empty(M, N // 2)
    a = ops.make_igemm_args(M, N, K, None, K, w2, None, N // 2, M, tf=L.TF_LAYERNORM_FOLDED, row_stats=st, ln_colsum=cs, bias=b2,
                            epi=L.EPI_GEGLU, tile_cfg=cfg, splitk=1, a_ps=xps, w_ps=ops.pack_wps(w2), out_ps=ops_ps2)
    ops.igemm(a)
    nb = (M // 32) * 32          # (rows of a ragged last block are padding)
    assert torch.equal(ops_ps2.view(-1)[: nb // 32 * (N // 32) * 3072], ops_ps.view(-1)[: nb // 32 * (N // 32) * 3072])


@pytest.mark.parametrize("cfg", [23, 24, 29])
def test_ps_gemm_batched_planes(ops, cfg):
    """A batch of independent problems (the 16 Winograd planes / 4 upsampling phases): blockIdx.z walks the PS buffers."""
    B, M, K, N = 4, 512, 320, 320
    x, w = rnd(620, B, M, K), rnd(621, B, K, N) / np.sqrt(K)
    wc = w.cuda().contiguous()
    ops.pack_wsplit(wc, batch=B)
    ref = torch.empty(B, M, N, device="cuda")
    a = ops.make_igemm_args(M, N, K, x.cuda(), K, wc, ref, N, M, batch=B, a_bstride=M * K, w_bstride=K * N, out_bstride=M * N,
                            tile_cfg=5, splitk=1, compute=2)
    ops.igemm(a)
    out = torch.empty(B, M, N, device="cuda")
    a = ops.make_igemm_args(M, N, K, None, K, wc, out, N, M, batch=B, out_bstride=M * N, tile_cfg=cfg, splitk=1,
                            a_ps=ops.pack_ps(x.cuda()), w_ps=ops.pack_wps(wc, batch=B))
    ops.igemm(a)
    assert torch.equal(out, ref)


def test_ps_gemm_rejections(ops):
    from dsml_thesis_amd import lib as L
    M, K, N = 256, 160, 160
    x, wp = rnd(630, M, K).cuda(), ops.pack_linear((rnd(631, N, K) / 12).cuda())
    out = torch.empty(M, N, device="cuda")
    ops.pack_wsplit(wp)
    with pytest.raises(L.LdmkError, match="a_ps / w_ps"):
        ops.igemm(ops.make_igemm_args(M, N, K, x, K, wp, out, N, M, tile_cfg=23, splitk=1, compute=2))
    xps, wps = ops.pack_ps(x), ops.pack_wps(wp)
    a = ops.make_igemm_args(M, N, K, None, K, wp, out, N, M, tile_cfg=23, splitk=1, a_ps=xps, w_ps=wps, epi=L.EPI_GEGLU)
    with pytest.raises(L.LdmkError, match="GEGLU"):
        ops.igemm(a)
    a = ops.make_igemm_args(M, N, K, None, K, wp, out, N, M, tile_cfg=23, splitk=2, a_ps=xps, w_ps=wps, out_ps=ops.ps_empty(M, N),
                            splitk_ws=torch.empty(2 * M * N, device="cuda"))
    with pytest.raises(L.LdmkError, match="out_ps"):
        ops.igemm(a)


@pytest.mark.parametrize("case", [(2, 320, 0, 16, 16), (1, 160, 160, 8, 8), (3, 64, 32, 4, 6), (1, 640, 0, 32, 32)])
def test_winograd_input_in_the_ps_layout(ops, case):
    """ldmk_winograd_input_ps: the 16 position planes of V = B^T d B (GroupNorm scale / shift + SiLU applied, two-source channel
    concat, zero padding of the activated tensor) written pre-split -- plane by plane exactly pack_ps of what ldmk_winograd_input
    writes."""
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
    Vps = ops.ps_empty(tiles, C, batch=16)
    L.call("ldmk_winograd_input_ps", x0.data_ptr(), c0, 0 if x1 is None else x1.data_ptr(), c1, coef.data_ptr(), 1, n, h, w,
           Vps.data_ptr(), ops.stream())
    ref = ops.pack_ps(V)
    nb = (tiles // 32) * (C // 16) * 3072          # whole row blocks (the rows of a ragged last block are padding)
    assert torch.equal(Vps[:, :nb], ref[:, :nb])
    for p in (0, 7, 15):
        hi, mid, lo = ops.unpack_ps(Vps[p].cpu(), tiles, C)
        assert torch.equal((hi.double() + mid.double() + lo.double()).float(), V[p].cpu())


@pytest.mark.parametrize("case", [(2, 320, 16, 16), (1, 64, 5, 7), (1, 640, 16, 16)])
def test_upconv_gather_in_the_ps_layout(ops, case):
    from dsml_thesis_amd import lib as L
    n, c, h, w = case
    x = rnd(710, n, h, w, c).cuda()
    pix = n * h * w
    A = torch.empty(4, pix, 4 * c, device="cuda")
    L.call("ldmk_upconv_gather", x.data_ptr(), c, n, h, w, A.data_ptr(), ops.stream())
    Aps = ops.ps_empty(pix, 4 * c, batch=4)
    L.call("ldmk_upconv_gather_ps", x.data_ptr(), c, n, h, w, Aps.data_ptr(), ops.stream())
    for p in range(4):
        hi, mid, lo = ops.unpack_ps(Aps[p].cpu(), pix, 4 * c)
        assert torch.equal(hi, _split3(A[p].cpu())[0]) and torch.equal((hi.double() + mid.double() + lo.double()).float(), A[p].cpu())
