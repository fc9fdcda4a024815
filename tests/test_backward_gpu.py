"""Op-level parity of the training-step kernels against PyTorch autograd on the CPU (float64
reference of the same op).  Tolerances are fp32 accumulation bounds, stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _autograd_on():
    """The reference side of these tests is autograd; other test modules switch it off process-wide."""
    with torch.enable_grad():
        yield


def _dev():
    return torch.device("cuda", 0)


def _rand(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)


def _close(got, ref, rtol, what):
    ref = ref.to(torch.float64)
    err = (got.detach().cpu().to(torch.float64) - ref).abs().max().item()
    bound = rtol * max(ref.abs().max().item(), 1e-30)
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e}"


@pytest.mark.parametrize("R,K,N,pad", [(4096, 160, 320, 0), (200, 64, 36, 8), (8192, 640, 160, 0), (1024, 320, 960, 0)])
def test_wgrad_rows(R, K, N, pad):
    from dsml_thesis_amd import train_ops as T
    a = _rand(R, K + pad, seed=1)
    dy = _rand(R, N, seed=2)
    ref = a[:, :K].double().t() @ dy.double()
    ad, dyd = a.to(_dev()), dy.to(_dev())
    dw = T.wgrad_linear(ad[:, :K], dyd)
    _close(dw, ref, 2e-5, "wgrad rows")
    dw1 = T.wgrad_linear(ad[:, :K], dyd, splitr=1)
    _close(dw1, ref, 2e-5, "wgrad rows, no split")
    T.wgrad_linear(ad[:, :K], dyd, dw=dw, accumulate=True)
    _close(dw, 2 * ref, 2e-5, "wgrad rows accumulate")
    again = T.wgrad_linear(ad[:, :K], dyd)
    assert torch.equal(again, T.wgrad_linear(ad[:, :K], dyd)), "wgrad must be bitwise reproducible"
    # bias gradient (column sums of dy) from the same pass, with and without a row split
    for sr in (0, 1):
        db = torch.empty(N, device=_dev())
        dwb = T.wgrad_linear(ad[:, :K], dyd, dbias=db, splitr=sr)
        _close(db, dy.double().sum(0), 2e-5, "fused bias gradient")
        _close(dwb, ref, 2e-5, "wgrad with fused bias gradient")


def test_wgrad_batched_heads():
    """dV_h = P_h^T dO_h for every (sample, head): batch over blockIdx.z, N = 32."""
    from dsml_thesis_amd import train_ops as T
    Z, Tk = 6, 256
    p = torch.softmax(_rand(Z, Tk, Tk, seed=3), -1)
    do = _rand(Z, Tk, 32, seed=4)
    ref = p.double().transpose(1, 2) @ do.double()
    pd, dod = p.to(_dev()), do.to(_dev())
    out = torch.empty(Z, Tk, 32, device=_dev())
    w = T.wgrad_args(Tk, Tk, 32, pd, dod, out, batch=Z, a_bstride=Tk * Tk, dy_bstride=Tk * 32, dw_bstride=Tk * 32)
    sr, need = T.wgrad_workspace_elems(w)
    ws = torch.empty(max(need, 1), device=_dev())
    w.splitr, w.ws, w.ws_elems = sr, ws.data_ptr(), ws.numel()
    T.wgrad(w)
    _close(out, ref, 2e-5, "batched wgrad")


@pytest.mark.parametrize("cin,cout,h,stride,ups", [(64, 160, 8, 1, False), (160, 160, 16, 2, False), (96, 64, 5, 2, False),
                                                     (64, 96, 4, 1, True), (320, 320, 8, 1, False)])
def test_conv3x3_backward(cin, cout, h, stride, ups):
    """weight and data gradients of a pad-1 3x3 convolution (optionally on a nearest-x2 upsampled input)."""
    from dsml_thesis_amd import ops, train_ops as T
    n = 2
    x = _rand(n, cin, h, h, seed=5).double().requires_grad_(True)
    w = (_rand(cout, cin, 3, 3, seed=6) / np.sqrt(9 * cin)).double().requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    y = F.conv2d(xin, w, None, stride=stride, padding=1)
    dy = _rand(*y.shape, seed=7)
    y.backward(dy.double())
    oh = y.shape[2]
    xd = x.detach().float().permute(0, 2, 3, 1).contiguous().to(_dev())
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(_dev())
    # weight gradient, in the packed forward layout
    db = torch.empty(cout, device=_dev())
    dwp = T.wgrad_conv3x3(xd, dyd, stride=stride, upsample=ups, dbias=db)
    ref_dw = ops.pack_conv3x3(w.grad.float().to(_dev()))
    _close(dwp, ref_dw.cpu(), 3e-5, "conv wgrad")
    _close(db, dy.double().sum((0, 2, 3)), 3e-5, "conv bias gradient")
    # data gradient through the mirrored-tap weights
    wp = ops.pack_conv3x3(w.detach().float().to(_dev()))
    wd = T.pack_dgrad3x3(wp, cin, cout)
    hh = 2 * h if ups else h
    dx = T.conv3x3_dgrad(dyd, wd, (hh, hh), stride=stride)
    if ups:
        dx = T.sumpool2(dx)
    _close(dx.permute(0, 3, 1, 2), x.grad, 3e-5, "conv dgrad")


@pytest.mark.parametrize("c0,c1,hw,silu", [(160, 0, 64, True), (320, 160, 256, True), (64, 0, 1024, False), (96, 32, 100, True)])
def test_group_norm_backward(c0, c1, hw, silu):
    from dsml_thesis_amd import ops, train_ops as T, lib as L
    n, C = 2, c0 + c1
    x0 = _rand(n, hw, c0, seed=8).double().requires_grad_(True)
    x1 = _rand(n, hw, c1, seed=9).double().requires_grad_(True) if c1 else None
    gamma = (1 + 0.1 * _rand(C, seed=10)).double().requires_grad_(True)
    beta = (0.1 * _rand(C, seed=11)).double().requires_grad_(True)
    xc = x0 if x1 is None else torch.cat([x0, x1], -1)
    z = F.group_norm(xc.permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
    y = F.silu(z) if silu else z
    dy = _rand(n, hw, C, seed=12)
    y.backward(dy.double())
    dev = _dev()
    x0d = x0.detach().float().to(dev)
    x1d = None if x1 is None else x1.detach().float().to(dev)
    g32, b32 = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    chunks = L.load().ldmk_gn_chunks(hw)
    partial = torch.empty(n * chunks * C * 3, device=dev)
    coef = torch.empty(n, 2, C, device=dev)
    ops.gn_coef(x0d, x1d, n, hw, g32, b32, 1e-5, partial=partial, coef=coef)
    p1 = partial[n * chunks * c0 * 3:] if c1 else None
    mr = T.gn_group_stats(partial, c0, p1, c1, n, hw, 32, 1e-5)
    dx0, dx1, dg, db = T.gn_bwd(x0d, x1d, dy.to(dev), coef, mr, g32, n, hw, silu=silu)
    _close(dx0, x0.grad, 5e-5, "gn dx0")
    if c1:
        _close(dx1, x1.grad, 5e-5, "gn dx1")
    _close(dg, gamma.grad, 5e-5, "gn dgamma")
    _close(db, beta.grad, 5e-5, "gn dbeta")
    # accumulate flags
    dx0b, _, dgb, _ = T.gn_bwd(x0d, x1d, dy.to(dev), coef, mr, g32, n, hw, silu=silu, dx0=dx0.clone(), acc0=True,
                               dx1=None if dx1 is None else dx1.clone(), dgamma=dg.clone(), dbeta=db.clone(), acc_params=True)
    _close(dx0b, 2 * x0.grad, 5e-5, "gn dx0 accumulate")
    _close(dgb, 2 * gamma.grad, 5e-5, "gn dgamma accumulate")


@pytest.mark.parametrize("rows,c", [(1024, 160), (300, 640), (64, 1024), (130, 36)])
def test_layer_norm_forward_backward(rows, c):
    from dsml_thesis_amd import ops, train_ops as T
    x = _rand(rows, c, seed=13).double().requires_grad_(True)
    gamma = (1 + 0.1 * _rand(c, seed=14)).double().requires_grad_(True)
    beta = (0.1 * _rand(c, seed=15)).double().requires_grad_(True)
    y = F.layer_norm(x, (c,), gamma, beta, 1e-5)
    dy = _rand(rows, c, seed=16)
    y.backward(dy.double())
    dev = _dev()
    xd, g32, b32 = x.detach().float().to(dev), gamma.detach().float().to(dev), beta.detach().float().to(dev)
    stats = ops.ln_stats(xd)
    yd = T.ln_apply(xd, stats, g32, b32)
    _close(yd, y.detach(), 1e-5, "ln forward")
    dx, dg, db = T.ln_bwd(dy.to(dev), xd, stats, g32)
    _close(dx, x.grad, 3e-5, "ln dx")
    _close(dg, gamma.grad, 3e-5, "ln dgamma")
    _close(db, beta.grad, 3e-5, "ln dbeta")


def test_geglu_softmax_colsum_small_ops():
    from dsml_thesis_amd import train_ops as T
    dev = _dev()
    # GEGLU
    pre = _rand(200, 2 * 96, seed=17).double().requires_grad_(True)
    v, g = pre.chunk(2, dim=-1)
    f = v * F.gelu(g)
    df = _rand(200, 96, seed=18)
    f.backward(df.double())
    pd = pre.detach().float().to(dev)
    _close(T.geglu_fwd(pd), f.detach(), 1e-5, "geglu fwd")
    _close(T.geglu_bwd(pd, df.to(dev)), pre.grad, 2e-5, "geglu bwd")
    # softmax backward with the attention scale folded in
    s = _rand(77, 300, seed=19).double().requires_grad_(True)
    p = torch.softmax(s * 0.25, -1)
    dp = _rand(77, 300, seed=20)
    p.backward(dp.double())
    ds = T.softmax_bwd_rows_(p.detach().float().to(dev), dp.to(dev).clone(), 0.25)
    _close(ds, s.grad, 2e-5, "softmax bwd")
    # column sums: whole tensor (bias gradient) and per sample (timestep-embedding gradient)
    x = _rand(6 * 1000, 160, seed=21)
    _close(T.colsum(x.to(dev)), x.double().sum(0, keepdim=True), 1e-5, "colsum")
    _close(T.colsum(x.to(dev), rows_per_group=1000), x.double().view(6, 1000, 160).sum(1), 1e-5, "colsum per sample")
    out = torch.ones(1, 160, device=dev)
    T.colsum(x.to(dev), out=out, accumulate=True)
    _close(out, x.double().sum(0, keepdim=True) + 1, 1e-5, "colsum accumulate")
    # SiLU, axpy, nearest-upsample backward
    z = _rand(1000, seed=22).double().requires_grad_(True)
    F.silu(z).backward(torch.ones(1000, dtype=torch.float64))
    zd = z.detach().float().to(dev)
    _close(T.silu(zd), F.silu(z).detach(), 1e-6, "silu")
    _close(T.silu_bwd(zd, torch.ones(1000, device=dev)), z.grad, 1e-5, "silu bwd")
    y = torch.ones(1000, device=dev)
    _close(T.axpy_(y, zd, 0.5), 1 + 0.5 * z.detach(), 1e-6, "axpy")
    u = _rand(2, 8, 6, 32, seed=23)
    ref = u.double().view(2, 4, 2, 3, 2, 32).sum((2, 4))
    _close(T.sumpool2(u.to(dev)), ref, 1e-6, "sumpool2")


def test_q_sample_mse_adamw_ema():
    from dsml_thesis_amd import train_ops as T
    dev = _dev()
    n = 4
    x0, noise = _rand(n, 3, 8, 8, seed=24), _rand(n, 3, 8, 8, seed=25)
    t = torch.tensor([0, 10, 500, 999])
    ac = torch.linspace(0.999, 0.01, 1000)
    a, b = ac.sqrt(), (1 - ac).sqrt()
    ref = a[t].view(n, 1, 1, 1) * x0 + b[t].view(n, 1, 1, 1) * noise
    got = T.q_sample(x0.to(dev), noise.to(dev), t.to(dev), a.to(dev), b.to(dev))
    assert torch.equal(got.cpu(), ref), "q_sample is two fp32 multiplies and one add: must be exact"
    pred = _rand(n, 3, 8, 8, seed=26).double().requires_grad_(True)
    loss = F.mse_loss(pred, noise.double())
    loss.backward()
    l, dp = T.mse_grad(pred.detach().float().to(dev), noise.to(dev))
    _close(l, loss.detach().view(1), 1e-6, "mse loss")
    _close(dp, pred.grad, 1e-6, "mse grad")
    # three AdamW steps against torch.optim.AdamW (ddpm.py:1363-1385 uses its defaults + lr from the config)
    p = torch.nn.Parameter(_rand(5000, seed=27))
    opt = torch.optim.AdamW([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    pd = p.detach().clone().to(dev)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in range(1, 4):
        g = _rand(5000, seed=30 + step)
        p.grad = g.clone()
        opt.step()
        T.adamw_(pd, g.to(dev), m, v, 1e-3, (0.9, 0.999), 1e-8, 1e-2, step)
    _close(pd, p.detach(), 2e-6, "adamw")
    sh = _rand(5000, seed=40).to(dev)
    ref = sh.cpu() - 0.01 * (sh.cpu() - p.detach())
    T.ema_(sh, pd, 0.01)
    _close(sh, ref, 2e-6, "ema")


@pytest.mark.parametrize("n,tokens,heads", [(2, 64, 5), (1, 256, 2), (2, 100, 1), (1, 1024, 3), (1, 37, 2)])
def test_attention_backward(n, tokens, heads):
    """d(qkv) of softmax(QK^T/sqrt(32))V against autograd; ragged token counts included."""
    from dsml_thesis_amd import ops, train_ops as T
    C_ = heads * 32
    qkv = _rand(n * tokens, 3 * C_, seed=50).double().requires_grad_(True)
    q, k, v = qkv.view(n, tokens, 3, heads, 32).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1)
    att = (p @ v).permute(0, 2, 1, 3).reshape(n * tokens, C_)
    datt = _rand(n * tokens, C_, seed=51)
    att.backward(datt.double())
    qd = qkv.detach().float().to(_dev())
    _close(ops.attn_self(qd, n, tokens, heads), att.detach(), 2e-5, "attention forward")
    att_d, lse = T.attn_self_lse(qd, n, tokens, heads)
    _close(att_d, att.detach(), 2e-5, "attention forward (lse variant)")
    ref_lse = torch.logsumexp(q @ k.transpose(-1, -2) * 32 ** -0.5, -1).detach()      # [n][heads][tokens]
    _close(lse, ref_lse, 1e-5, "log-sum-exp")
    dq_flash = T.attn_self_bwd(qd, att_d, datt.to(_dev()), lse, n, tokens, heads)
    _close(dq_flash, qkv.grad, 3e-5, "flash attention backward")
    assert torch.equal(dq_flash, T.attn_self_bwd(qd, att_d, datt.to(_dev()), lse, n, tokens, heads)), "must be reproducible"
    if tokens % 32 == 0:               # the materialised variant (batched GEMMs) needs whole 32-token tiles
        dqkv = T.attention_backward(qd, datt.to(_dev()), n, tokens, heads)
        _close(dqkv, qkv.grad, 3e-5, "materialised attention backward")


@pytest.mark.parametrize("n,tokens,L_ctx,heads", [(2, 64, 3, 5), (1, 100, 77, 2), (2, 16, 1, 1)])
def test_cross_attention_backward(n, tokens, L_ctx, heads):
    from dsml_thesis_amd import ops, train_ops as T
    C_ = heads * 32
    q = _rand(n * tokens, C_, seed=60).double().requires_grad_(True)
    k = _rand(n * L_ctx, C_, seed=61).double().requires_grad_(True)
    v = _rand(n * L_ctx, C_, seed=62).double().requires_grad_(True)
    qh = q.view(n, tokens, heads, 32).permute(0, 2, 1, 3)
    kh = k.view(n, L_ctx, heads, 32).permute(0, 2, 1, 3)
    vh = v.view(n, L_ctx, heads, 32).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * 32 ** -0.5, -1)
    out = (p @ vh).permute(0, 2, 1, 3).reshape(n * tokens, C_)
    dout = _rand(n * tokens, C_, seed=63)
    out.backward(dout.double())
    dev = _dev()
    qd, kd, vd = q.detach().float().to(dev), k.detach().float().to(dev), v.detach().float().to(dev)
    _close(ops.attn_cross(qd, kd, vd, n, tokens, L_ctx, heads), out.detach(), 2e-5, "cross attention forward")
    dq, dk, dv = T.attn_cross_bwd(qd, kd, vd, dout.to(dev), n, tokens, L_ctx, heads)
    _close(dq, q.grad, 3e-5, "cross attention dq")
    _close(dk, k.grad, 3e-5, "cross attention dk")
    _close(dv, v.grad, 3e-5, "cross attention dv")


def test_audio_attention_backward():
    """Conv1DTemporalAttention (9-/17-frame wav2vec2 window -> pooled feature): parameter gradients of the fused HIP
    kernel pair against autograd through the same nn.Modules on the CPU (float64)."""
    from dsml_thesis_amd.encoders import Conv1DTemporalAttention, audio_attention_backward
    from dsml_thesis_amd.synth import load_recipe
    for T_win in (17, 9):
        mod = Conv1DTemporalAttention(seq_len=T_win, subspace_dim=768)
        load_recipe(mod)
        x = _rand(3, T_win, 768, seed=70)
        dout = _rand(3, 1, 768, seed=71)
        ref = Conv1DTemporalAttention(seq_len=T_win, subspace_dim=768).double()
        ref.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
        xt = x.double().transpose(1, 2)
        attn = ref.attentionNet(ref.attentionConvNet(xt).view(3, T_win)).view(3, T_win, 1)
        out = torch.bmm(xt, attn).view(3, 768).unsqueeze(1)
        out.backward(dout.double())
        mod = mod.cuda()
        _close(mod(x.cuda()), out.detach(), 2e-5, "audio attention forward")
        audio_attention_backward(mod, x.cuda(), dout.cuda())
        for (k, p), (_, pr) in zip(mod.named_parameters(), ref.named_parameters()):
            _close(p.grad, pr.grad, 5e-5, f"audio attention grad {k}")

