"""GPU: bf16 matrix-core compute of the training-step GEMMs (BASELINE configs[4]: forward / data-gradient igemm and the
weight-gradient GEMM with LDMK_COMPUTE_BF16).  HBM tensors, accumulation, prologues and epilogues stay fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rnd
from test_ops_gpu import close, nchw, nhwc, ops  # noqa: F401  (the `ops` fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _autograd_on():
    """Reference sides use autograd; other test modules switch it off process-wide."""
    with torch.enable_grad():
        yield

# bf16 matrix-core compute (BASELINE configs[4]).  Tolerance, stated: both operands of every product are rounded to bf16
# (8 significand bits, relative rounding error <= 2^-9 each), products and sums are fp32, so an output element differs
# from the fp32 result by at most ~2^-8 * sum|a_k b_k|; against the typical |sum a_k b_k| ~ sqrt(K) * |a||b| of these
# random operands that is a few 1e-3 relative.  The exact check: feeding operands that ARE bf16-representable must
# reproduce the fp32 kernel to fp32 summation-order accuracy.
def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("b_trans", [False, True])
@pytest.mark.parametrize("M,K,N", [(300, 320, 160), (128, 640, 1920), (4096, 160, 480)])
def test_bf16_linear_forward_and_dgrad_form(ops, M, K, N, b_trans):
    from dsml_thesis_amd import lib as L
    x, w, b = rnd(400, M, K), rnd(401, N, K) / np.sqrt(K), 0.1 * rnd(402, N)
    res = rnd(403, M, N)
    wp = w.cuda() if b_trans else ops.pack_linear(w.cuda())
    y = ops.linear(x.cuda(), wp, b.cuda(), residual=res.cuda(), b_trans=b_trans, compute=L.COMPUTE_BF16)
    exact = _bf16_round(x).double() @ _bf16_round(w).double().t() + b.double() + res.double()
    close(y, exact.float(), 2e-5, 2e-5)                          # what the kernel computes, to fp32 accumulation accuracy
    ref = x.double() @ w.double().t() + b.double() + res.double()
    assert (y.cpu().double() - ref).abs().max() < 8e-3 * (x.abs().max() * w.abs().max() * np.sqrt(K)).item()


@pytest.mark.parametrize("case", [(2, 160, 320, 16, 16, 1, False), (2, 64, 96, 9, 7, 1, False), (1, 160, 160, 16, 16, 2, False),
                                  (1, 160, 160, 8, 8, 1, True), (3, 640, 640, 8, 8, 1, False)])
def test_bf16_conv3x3(ops, case):
    from dsml_thesis_amd import lib as L
    n, cin, cout, h, w, stride, up = case
    x, wt, b = rnd(410, n, cin, h, w), rnd(411, cout, cin, 3, 3) / np.sqrt(9 * cin), 0.1 * rnd(412, cout)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    y = ops.conv3x3(nhwc(x), ops.pack_conv3x3(wt.cuda()), b.cuda(), stride=stride, upsample=up, compute=L.COMPUTE_BF16)
    exact = F.conv2d(_bf16_round(xi).double(), _bf16_round(wt).double(), b.double(), stride=stride, padding=1)
    close(nchw(y), exact.float(), 2e-5, 2e-5)


@pytest.mark.parametrize("conv", [False, True])
def test_bf16_wgrad(ops, conv):
    from dsml_thesis_amd import lib as L
    from dsml_thesis_amd import train_ops as T
    T.set_compute("bf16")
    try:
        if conv:
            n, c, co, h, w = 2, 160, 320, 16, 16
            x, dy = rnd(420, n, c, h, w), rnd(421, n, co, h, w)
            dw = torch.zeros(9 * c, co, device="cuda")
            db = torch.zeros(co, device="cuda")
            T.wgrad_conv3x3(nhwc(x), nhwc(dy), dw=dw, dbias=db)
            xr, dyr = _bf16_round(x).double().requires_grad_(False), _bf16_round(dy).double()
            wt = torch.zeros(co, c, 3, 3, dtype=torch.double, requires_grad=True)
            (F.conv2d(xr, wt, padding=1) * dyr).sum().backward()
            ref = ops.pack_conv3x3(wt.grad.float().cuda())
            close(dw, ref, 5e-5, 5e-4)
            close(db, dy.double().sum((0, 2, 3)).float(), 1e-5, 1e-4)            # bias sums stay fp32
        else:
            R, K, N = 5000, 320, 160
            a, dy = rnd(422, R, K), rnd(423, R, N)
            dw = torch.zeros(K, N, device="cuda")
            db = torch.zeros(N, device="cuda")
            T.wgrad_linear(a.cuda(), dy.cuda(), dw=dw, dbias=db)
            close(dw, (_bf16_round(a).double().t() @ _bf16_round(dy).double()).float(), 5e-5, 5e-4)
            close(db, dy.double().sum(0).float(), 1e-5, 1e-4)
    finally:
        T.set_compute("f32")


@pytest.mark.parametrize("n,tokens,heads", [(2, 256, 5), (1, 1024, 5), (3, 100, 3), (1, 64, 20)])
def test_bf16_attention_forward_and_backward(n, tokens, heads):
    """Self attention of the bf16 training step (csrc/attention_bf16.hip): Q K^T, P V and the five backward products on the
    bf16 matrix cores, fp32 softmax / statistics / storage.  Tolerance, stated: output and gradients within 1.5 % relative
    L2 of float64 autograd on the fp32 inputs (bf16 operand rounding 2^-9 on Q, K, V, dO and on the probabilities; measured
    ~0.3-0.6 %); and within 2e-3 of the fp32 kernels' log-sum-exp."""
    from dsml_thesis_amd import train_ops as T
    C = heads * 32
    qkv = (rnd(430, n * tokens, 3 * C) * 0.8).double().requires_grad_(True)
    dout = rnd(431, n * tokens, C).double()
    q, k, v = (t_.reshape(n, tokens, heads, 32).permute(0, 2, 1, 3) for t_ in qkv.chunk(3, dim=1))
    sc = q @ k.transpose(-1, -2) * 32 ** -0.5
    att = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(n * tokens, C)
    (att * dout).sum().backward()
    lse_ref = torch.logsumexp(sc, dim=-1)                                   # [n][heads][tokens]
    qd, dd = qkv.detach().float().cuda(), dout.float().cuda()
    T.set_compute("bf16")
    try:
        out, lse = T.attn_self_lse(qd, n, tokens, heads)
        dqkv = T.attn_self_bwd(qd, out, dd, lse, n, tokens, heads)
    finally:
        T.set_compute("f32")
    rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
    e_out, e_lse, e_grad = rel(out, att.detach()), (lse.double().cpu() - lse_ref.detach()).abs().max().item(), rel(dqkv, qkv.grad)
    print(f"bf16 attention n={n} tokens={tokens} heads={heads}: out {e_out:.2e}, lse {e_lse:.2e}, dqkv {e_grad:.2e}")
    assert e_out < 1.5e-2 and e_lse < 2e-2 and e_grad < 1.5e-2
    for part, name in zip(dqkv.double().cpu().chunk(3, dim=1), "qkv"):
        ref = qkv.grad.chunk(3, dim=1)["qkv".index(name)]
        assert ((part - ref).norm() / ref.norm()).item() < 2e-2, name
    out32, lse32 = T.attn_self_lse(qd, n, tokens, heads)                     # the fp32 kernels are untouched by the mode switch
    assert rel(out32, att.detach()) < 1e-5


def test_bf16_training_step_gradients_against_float64_autograd():
    """BASELINE configs[4]: p_losses forward + backward with every GEMM on the bf16 matrix cores.  Tolerance, stated: per
    parameter tensor, the relative L2 error of the gradient vs float64 autograd on the oracle is below 3 % (bf16 operand
    rounding, 2^-9 per operand, propagated through ~20 layers forward and back; measured worst 1.3 %), the loss within
    0.5 %; and the step still trains (three AdamW steps lower the loss)."""
    with torch.enable_grad():
        from test_train_gpu import SMALL, _oracle_grads, _setup
        from dsml_thesis_amd.train import UNetTrainer, reference_grad_layout
        m, _, sd, x0, noise, ctx, t = _setup(SMALL, 2, 16)
        tr = UNetTrainer(m, compute="bf16")
        loss_ref, _, grads, _, sched = _oracle_grads(SMALL, sd, x0, noise, ctx, t)
        sa, sb = sched["sqrt_alphas_cumprod"].cuda(), sched["sqrt_one_minus_alphas_cumprod"].cuda()
        loss = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sa, sb)
        assert abs(loss.item() - loss_ref.item()) <= 5e-3 * abs(loss_ref.item()), (loss.item(), loss_ref.item())
        gdev = {k: (torch.zeros_like(sd[k]) if v is None else v.float()).cuda() for k, v in grads.items()}
        worst = (0.0, "")
        for name, g in tr.P.g.items():
            ref = reference_grad_layout(m, name, gdev).double().cpu()
            nrm = ref.norm().item()
            if nrm < 1e-12:
                assert g.abs().max().item() == 0.0, name          # dead branches stay exactly zero
                continue
            err = (g.double().cpu() - ref).norm().item() / nrm
            worst = max(worst, (err, name))
            assert err <= 3e-2, f"bf16 gradient {name}: relative L2 error {err:.3e}"
        print("worst bf16 gradient error", worst)
        losses = [loss.item()]
        for _ in range(3):
            tr.adamw_step(lr=2e-5)
            losses.append(tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sa, sb).item())
        assert losses[-1] < losses[0], losses
        # the fp32 trainer on the same module is unaffected by the bf16 one (the mode is per trainer)
        tr32 = UNetTrainer(m)
        l32 = tr32.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sa, sb)
        assert tr32.compute == 0 and tr.compute == 1 and np.isfinite(l32.item())


def test_latent_diffusion_trains_in_bf16_through_the_facade():
    """`model.train_compute = "bf16"` (the reference: Trainer(precision=...)): training_step_latents runs the bf16 GEMMs,
    updates UNet + conditioner + EMA, and the loss goes down."""
    from helpers import make_fr_model
    model = make_fr_model(gain=0.5).train()
    model.train_compute = "bf16"
    model.cond_stage_model.p_uncond = 0.0
    z = rnd(90, 2, 3, 32, 32).cuda()
    batch = {"class_label": torch.tensor([1, 5]).cuda()}
    losses = []
    for _ in range(3):
        loss, _ = model.training_step_latents(z, batch, lr=1e-5, t=torch.tensor([300, 800]).cuda(),
                                              noise=rnd(93, 2, 3, 32, 32).cuda())
        losses.append(loss.item())
    assert model.trainer().compute == 1 and all(np.isfinite(losses)) and losses[2] < losses[0], losses


@pytest.mark.parametrize("cfg", [1, 2, 4, 5])
@pytest.mark.parametrize("M,K,N", [(300, 320, 160), (1024, 640, 1920), (4096, 160, 480)])
def test_bf16_forward_gemm_with_prepacked_weight_image_is_bitwise_the_in_kernel_conversion(ops, M, K, N, cfg):
    """The training step packs its forward weights once per optimiser step (bf16, transposed, K-contiguous: ldmk_pack_wbf16t);
    the GEMM then copies the image instead of converting fp32 W fragment by fragment.  Same rounding, same products."""
    from dsml_thesis_amd import lib as L
    x, w, b = rnd(450, M, K), rnd(451, N, K) / np.sqrt(K), 0.1 * rnd(452, N)
    wp = ops.pack_linear(w.cuda())
    img = ops.pack_wbf16t(wp)
    assert torch.equal(img[:, :K].float(), wp.t().to(torch.bfloat16).float()) and torch.count_nonzero(img[:, K:]).item() == 0
    outs = []
    for packed in (None, img):
        out = torch.empty(M, N, device="cuda")
        a = ops.make_igemm_args(M, N, K, x.cuda(), K, wp, out, N, M, bias=b.cuda(), tile_cfg=cfg, splitk=1, compute=L.COMPUTE_BF16,
                                w_bf16t=packed)
        ops.igemm(a)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def test_bf16_weight_images_belong_to_their_trainer():
    """The pre-packed bf16 forward-weight images are registered per trainer (train.Bf16Images), only for weights inside that
    trainer's flat parameter buffer.  Build one bf16 trainer, run it, free it and return its memory to the driver; a second
    trainer then repacks ONLY its own weights (the first one's addresses are gone) and computes the loss of a fresh trainer."""
    import gc
    with torch.enable_grad():
        from test_train_gpu import SMALL, _setup
        from dsml_thesis_amd import train as TR
        m, _, sd, x0, noise, ctx, t = _setup(SMALL, 2, 16)
        from oracle import ldm_oracle as O
        from oracle import weights as W
        sched = O.register_schedule(**W.SCHEDULE)
        sa, sb = sched["sqrt_alphas_cumprod"].cuda(), sched["sqrt_one_minus_alphas_cumprod"].cuda()
        args = (x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sa, sb)
        tr1 = TR.UNetTrainer(m, compute="bf16")
        l_first = tr1.p_losses(*args).item()          # registers the images
        l1 = tr1.p_losses(*args).item()               # uses them
        assert l_first == l1, "pre-packed images reproduce the in-kernel conversion bit for bit"
        reg1 = tr1._wt16
        assert reg1.images and all(reg1.owns(p) for p in reg1.images)
        n_images = len(reg1.images)
        del tr1, reg1
        gc.collect()
        torch.cuda.empty_cache()                      # the first trainer's parameter segment goes back to the driver
        tr2 = TR.UNetTrainer(m, compute="bf16")
        assert not tr2._wt16.images
        l2a = tr2.p_losses(*args).item()
        l2b = tr2.p_losses(*args).item()
        assert TR._WT16 is tr2._wt16 and len(tr2._wt16.images) == n_images
        assert l2a == l2b == l1
        # two live trainers do not repack each other's weights
        tr3 = TR.UNetTrainer(m, compute="bf16")
        tr3.p_losses(*args)
        assert set(tr3._wt16.images).isdisjoint(tr2._wt16.images)
        assert tr2.p_losses(*args).item() == l1
