"""GPU: the conv-mode pre-split tile (csrc/igemm_ps.hip: igemm_psc_kernel; tile_cfg 23 / 24 / 26 / 27 with a_mode = LDMK_A_CONV3X3)
and its producer ldmk_gn_apply_ps_h2.  The 3x3 convolution's A operand is the GroupNorm-applied activation stored ONCE in the F16X2
PS layout; the nine taps are per-lane LDS-DMA addresses into it.  The bar is the one of every pre-split tile: the SAME bits as the
LDS-tiled F16X2 implicit GEMM (tile_cfg 5 / 1) on the fp32 ldmk_gn_apply output at equal split-K -- same split values, same products
in the same order -- so the accuracy statements of tests/test_f16x2_gpu.py (float64 reference, the reference fixtures through the
UNet) carry over unchanged."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rnd
from test_ops_gpu import close, nchw, nhwc, ops  # noqa: F401  (the `ops` fixture)

pytestmark = pytest.mark.gpu


def _flag():
    return torch.zeros(1, dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("case", [(2, 160, 0, 16, 16), (3, 64, 0, 8, 8), (1, 32, 32, 64, 64), (2, 96, 64, 32, 32), (5, 320, 0, 8, 8)])
def test_gn_apply_ps_h2_is_pack_ps_of_gn_apply(ops, case):
    """The producer: GroupNorm scale / shift + SiLU of the two-source channel concat, written in the PS layout == ldmk_pack_ps_h2 of
    ldmk_gn_apply's fp32 output, bit for bit (ragged last row block included); an element of 1000 or more raises the flag."""
    n, c0, c1, h, w = case
    x0 = nhwc(rnd(700, n, c0, h, w)).contiguous()
    x1 = nhwc(rnd(701, n, c1, h, w)).contiguous() if c1 else None
    C = c0 + c1
    coef = torch.stack([1.0 + 0.2 * rnd(702, n, C), 0.3 * rnd(703, n, C)], 1).contiguous().cuda()
    flag = _flag()
    y = ops.gn_apply(x0, x1, coef, n, h * w, silu=True)
    yps = ops.gn_apply_ps(x0, x1, coef, n, h * w, flag, silu=True)
    assert torch.equal(yps, ops.pack_ps(y, h2_flag=flag)) and int(flag.item()) == 0
    y2 = ops.gn_apply(x0, x1, coef, n, h * w, silu=False)
    assert torch.equal(ops.gn_apply_ps(x0, x1, coef, n, h * w, flag, silu=False), ops.pack_ps(y2, h2_flag=flag))
    coef[0, 1, 3] = 5000.0
    ops.gn_apply_ps(x0, x1, coef, n, h * w, flag, silu=False)
    assert int(flag.item()) == 1


CASES = [  # n, cin, cout, h, w, stride, splitk
    (2, 160, 160, 16, 16, 1, 1), (1, 160, 160, 64, 64, 1, 1), (3, 64, 96, 8, 8, 1, 1), (2, 64, 320, 9, 7, 1, 2), (2, 320, 640, 8, 8, 1, 5),
    (1, 160, 320, 16, 16, 2, 1), (2, 96, 160, 32, 32, 1, 3), (1, 640, 640, 16, 16, 1, 4)]


@pytest.mark.parametrize("cfg", [23, 24, 26, 27])
@pytest.mark.parametrize("case", CASES)
def test_conv_mode_ps_tile_is_bitwise_the_lds_tiled_f16x2_convolution(ops, case, cfg):
    """bias + per-sample vector + residual epilogue, ragged M, image widths that are not multiples of the 32-row block, halo on all
    four sides, stride 2, K split on chunk boundaries -- against tile_cfg 5 in F16X2 reading the fp32 tensor (which scales and splits
    every element nine times per column tile)."""
    from dsml_thesis_amd import lib as L
    n, cin, cout, h, w, stride, sk = case
    x = nhwc(rnd(710, n, cin, h, w)).contiguous()
    wt, b = rnd(711, cout, cin, 3, 3) / np.sqrt(9 * cin), 0.1 * rnd(712, cout)
    coef = torch.stack([1.0 + 0.2 * rnd(713, n, cin), 0.3 * rnd(714, n, cin)], 1).contiguous().cuda()
    wp = ops.pack_conv3x3(wt.cuda())
    ops.pack_wsplit_h2(wp)
    wps = ops.pack_wps(wp, h2=True)
    flag = _flag()
    y = ops.gn_apply(x, None, coef, n, h * w, silu=True).view(n, h, w, cin)
    yps = ops.gn_apply_ps(x, None, coef, n, h * w, flag, silu=True)
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    M = n * oh * ow
    res, vec = rnd(715, M, cout).cuda(), rnd(716, n, cout).cuda()
    ws = torch.empty(8 * M * cout, device="cuda")
    ref = torch.empty(n, oh, ow, cout, device="cuda")
    a = ops.make_igemm_args(M, cout, 9 * cin, y, cin, wp, ref, cout, oh * ow, conv=(h, w, oh, ow, stride, 1, 0), bias=b.cuda(), residual=res,
                            batch_vec=vec, batch_vec_ld=cout, tile_cfg=5, splitk=sk, splitk_ws=ws, compute=L.COMPUTE_F16X2, range_flag=flag)
    ops.igemm(a)
    out = ops.conv3x3_ps(yps, n, h, w, cin, wp, wps, flag, bias=b.cuda(), stride=stride, batch_vec=vec, residual=res, tile_cfg=cfg, splitk=sk,
                         splitk_ws=ws)
    assert int(flag.item()) == 0
    assert torch.equal(out, ref)
    # ... and the convolution it is: float64 F.conv2d of the normalised input
    xa = F.silu(nchw(x).double().cpu() * coef[:, 0].double().cpu()[:, :, None, None] + coef[:, 1].double().cpu()[:, :, None, None])
    r64 = F.conv2d(xa, wt.double(), b.double(), stride=stride, padding=1) + nchw(res.view(n, oh, ow, cout)).double().cpu() + vec.double().cpu()[:, :, None, None]
    close(nchw(out), r64.float(), 2e-5, 2e-5)
    if M % 32 == 0 and (oh * ow) % 32 == 0 and sk == 1:      # GroupNorm records of the result: the lane = column form of the kernel
        rec_ref, o2 = torch.zeros(M // 32, cout, 3, device="cuda"), torch.empty(n, oh, ow, cout, device="cuda")
        a = ops.make_igemm_args(M, cout, 9 * cin, y, cin, wp, o2, cout, oh * ow, conv=(h, w, oh, ow, stride, 1, 0), bias=b.cuda(), tile_cfg=5,
                                splitk=1, compute=L.COMPUTE_F16X2, range_flag=flag)
        a.stats_out = rec_ref.data_ptr()
        ops.igemm(a)
        rec = torch.zeros(M // 32, cout, 3, device="cuda")
        o3 = ops.conv3x3_ps(yps, n, h, w, cin, wp, wps, flag, bias=b.cuda(), stride=stride, tile_cfg=cfg, stats_out=rec)
        assert torch.equal(o3, o2) and torch.equal(rec, rec_ref)


def test_conv_mode_rejections(ops):
    from dsml_thesis_amd import lib as L
    n, cin, cout, h, w = 1, 64, 64, 8, 8
    wp = ops.pack_conv3x3((rnd(720, cout, cin, 3, 3) / 24.0).cuda())
    wps, wps3 = ops.pack_wps(wp, h2=True), ops.pack_wps(wp)
    flag = _flag()
    yps = ops.pack_ps(rnd(721, n * h * w, cin).cuda(), h2_flag=flag)
    ops.conv3x3_ps(yps, n, h, w, cin, wp, wps, flag, tile_cfg=27)                           # fine
    with pytest.raises(L.LdmkError, match="divide"):
        ops.conv3x3_ps(yps, n, h, w, cin, wp, wps, flag, tile_cfg=27, splitk=3, splitk_ws=torch.empty(1 << 20, device="cuda"))
    with pytest.raises(L.LdmkError, match="conv mode"):
        ops.conv3x3_ps(yps, n, h, w, cin, wp, wps, flag, tile_cfg=25)                       # a GEGLU-pair tile
    with pytest.raises(L.LdmkError, match="f16x2"):
        ops.conv3x3_ps(ops.pack_ps(rnd(721, n * h * w, cin).cuda()), n, h, w, cin, wp, wps3, None, tile_cfg=27)      # bf16x3 planes


def test_unet_with_every_eligible_convolution_on_the_conv_mode_tile(monkeypatch):
    """UNetModel.forward with LDMK_PSC_FORCE: every ResBlock convolution whose channels allow it runs as a direct convolution on the
    conv-mode pre-split tile (no Winograd, no in-register split) -- eps against the reference's fixture at the unchanged 3e-5."""
    from conftest import golden
    from oracle import weights as W
    from dsml_thesis_amd.unet import UNetModel
    monkeypatch.setenv("LDMK_PSC_FORCE", "27,1")
    g = golden("g4_unet_fr.npz")
    m = UNetModel(**W.FR_UNET)
    m.load_state_dict(W.synth_state_dict(W.unet_param_shapes(W.FR_UNET)), strict=True)
    m = m.cuda().eval()
    m.policy_batch = 16
    x, t, ctx = rnd(41, 2, 3, 32, 32), torch.tensor([3, 981]), rnd(42, 2, 1, 512)       # (the inputs of the fixture: tests/test_unet_gpu.py)
    eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
    pg = m.program(x.shape[0], 32, 32, 1, 0)
    names = [c[3] for c in pg.calls]
    assert names.count("ldmk_gn_apply_ps_h2") >= 30 and "ldmk_winograd_input_ps_h2" not in names
    close(eps, g["fr_eps"], 3e-5, 3e-5)
    d = float((eps.cpu() - torch.from_numpy(g["fr_eps"])).abs().max())
    print(f"UNet eps, every ResBlock convolution on the conv-mode pre-split tile: max |diff| vs the reference {d:.3e}")
