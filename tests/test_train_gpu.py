"""GPU: the UNet training step -- forward, loss and every parameter gradient of
`p_losses` against PyTorch autograd run on the oracle (CPU, float64 weights of the same recipe)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, rnd
from oracle import ldm_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _autograd_on():
    """The reference side of these tests is autograd; other test modules switch it off process-wide."""
    with torch.enable_grad():
        yield

SMALL = dict(W.FR_UNET, model_channels=64, channel_mult=[1, 2], num_res_blocks=1, attention_resolutions=[2, 1])


def _setup(cfg, n, hw, seed=0, gain=1.0):
    from dsml_thesis_amd.unet import UNetModel
    from dsml_thesis_amd.train import UNetTrainer
    m = UNetModel(**cfg)
    sd = W.synth_state_dict(W.unet_param_shapes(cfg), gain=gain)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    tr = UNetTrainer(m)
    x0 = rnd(seed + 1, n, cfg["in_channels"], hw, hw)
    noise = rnd(seed + 2, n, cfg["out_channels"], hw, hw)
    ctx = rnd(seed + 3, n, 1, cfg["context_dim"])
    t = torch.tensor([17, 803, 400, 999][:n])
    return m, tr, sd, x0, noise, ctx, t


def _oracle_grads(cfg, sd, x0, noise, ctx, t, dtype=torch.float64):
    sched = O.register_schedule(**W.SCHEDULE)
    return _oracle_grads_inner(cfg, sd, x0, noise, ctx, t, dtype, sched)


def _oracle_grads_inner(cfg, sd, x0, noise, ctx, t, dtype, sched):
    sdg = {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}
    ctxg = ctx.to(dtype).requires_grad_(True)
    a = sched["sqrt_alphas_cumprod"][t].view(-1, 1, 1, 1)
    b = sched["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    x_noisy = (a * x0 + b * noise).to(dtype)          # q_sample in fp32 (ddpm.py:1009-1012), then promoted
    eps = O.unet_forward(sdg, cfg, x_noisy, t, ctxg)
    loss = F.mse_loss(eps, noise.to(dtype))           # get_loss 'l2' + mean (ddpm.py:324-334, 1034)
    loss.backward()
    return loss.detach(), eps.detach(), {k: v.grad for k, v in sdg.items()}, ctxg.grad, sched


def _check_all_grads(m, tr, grads, rtol):
    from dsml_thesis_amd.train import reference_grad_layout
    gdev = {k: (torch.zeros_like(sd_v, dtype=torch.float32) if v is None else v.float()).cuda()
            for (k, v), sd_v in zip(grads.items(), grads.values())}
    worst = (0.0, "")
    for name, g in tr.P.g.items():
        ref = reference_grad_layout(m, name, gdev).double().cpu()
        got = g.double().cpu()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = max(ref.abs().max().item(), 1e-12)
        err = (got - ref).abs().max().item() / scale
        if err > worst[0]:
            worst = (err, name)
        assert err <= rtol, f"gradient {name}: relative max error {err:.3e} > {rtol}"
    return worst


def test_small_unet_p_losses_gradients():
    m, tr, sd, x0, noise, ctx, t = _setup(SMALL, 2, 16)
    loss_ref, eps_ref, grads, dctx_ref, sched = _oracle_grads(SMALL, sd, x0, noise, ctx, t)
    loss = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                       sched["sqrt_one_minus_alphas_cumprod"].cuda())
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item()), (loss.item(), loss_ref.item())
    # to_q / to_k / norm2 of the single-token cross-attention receive exactly zero gradient in the reference too
    for k, v in grads.items():
        if any(s in k for s in ("attn2.to_q", "attn2.to_k", "norm2.")):
            assert v is None or v.abs().max().item() == 0.0, k
    worst = _check_all_grads(m, tr, grads, 1e-4)      # fp32 kernels vs float64 autograd (measured 1.7e-5)
    print("worst gradient error", worst)
    err = (tr.dctx.double().cpu() - dctx_ref.view_as(tr.dctx.cpu())).abs().max().item() / dctx_ref.abs().max().item()
    assert err <= 2e-4, f"context gradient {err:.3e}"


def test_training_forward_matches_sampling_forward():
    """The training forward (materialised norms, padded boundary convolutions) and the sampling program compute the
    same function."""
    m, tr, sd, x0, noise, ctx, t = _setup(SMALL, 2, 16)
    eps_t = tr.forward(x0.cuda(), t.cuda(), ctx.cuda())
    eps_s = m(x0.cuda(), t.cuda(), context=ctx.cuda())
    torch.testing.assert_close(eps_t, eps_s, rtol=2e-4, atol=2e-5)


def test_full_fr_unet_gradients_and_adamw_step():
    """Shipped FR UNet (156.8 M parameters) at 32x32, batch 1: gradients vs autograd, then one AdamW step vs
    torch.optim.AdamW on the same gradients."""
    cfg = W.FR_UNET
    m, tr, sd, x0, noise, ctx, t = _setup(cfg, 1, 32)
    loss_ref, _, grads, _, sched = _oracle_grads(cfg, sd, x0, noise, ctx, t, dtype=torch.float32)
    loss = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                       sched["sqrt_one_minus_alphas_cumprod"].cuda())
    assert abs(loss.item() - loss_ref.item()) <= 5e-5 * abs(loss_ref.item())
    worst = _check_all_grads(m, tr, grads, 1e-4)      # both sides fp32 here, 60 layers (measured 1.9e-6)
    print("worst gradient error", worst)
    before = tr.P.flat.clone()
    g = tr.P.grad.clone()
    tr.adamw_step(lr=2e-6)        # Adam's first step moves every weight by ~lr: keep it inside the linear regime
    p = torch.nn.Parameter(before.clone())
    opt = torch.optim.AdamW([p], lr=2e-6)
    p.grad = g
    opt.step()
    torch.testing.assert_close(tr.P.flat, p.detach(), rtol=1e-6, atol=1e-7)
    loss2 = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                        sched["sqrt_one_minus_alphas_cumprod"].cuda())
    assert loss2.item() < loss.item(), "one AdamW step on the same batch must lower the loss"


def test_state_dict_round_trip_is_exact():
    """flat packed parameters -> reference state-dict layout reproduces the loaded weights bit for bit."""
    m, tr, sd, *_ = _setup(SMALL, 2, 16)
    back = tr.state_dict_reference()
    assert set(back) == set(sd)
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k


def test_latent_diffusion_training_step_updates_unet_conditioner_and_ema():
    from helpers import make_fr_model
    model = make_fr_model(gain=0.5).train()
    n = 2
    z = rnd(90, n, 3, 32, 32).cuda()
    batch = {"class_label": torch.tensor([1, 5]).cuda()}
    unet = model.model.diffusion_model
    x, t, ctx = rnd(91, n, 3, 32, 32).cuda(), torch.tensor([100, 700]).cuda(), rnd(92, n, 1, 512).cuda()
    eps0 = unet(x, t, context=ctx)
    emb0 = model.cond_stage_model.embedding.weight.detach().clone()
    model.cond_stage_model.p_uncond = 0.0            # keep the class tokens (the null-class draw is random)
    losses = []
    for _ in range(3):
        loss, ld = model.training_step_latents(z, batch, lr=1e-5, t=torch.tensor([300, 800]).cuda(), noise=rnd(93, n, 3, 32, 32).cuda())
        losses.append(loss.item())
        assert set(ld) == {"train_loss_simple", "train_loss"}
    assert all(np.isfinite(losses)) and losses[2] < losses[0], losses
    tr = model.trainer()
    assert not torch.equal(model.cond_stage_model.embedding.weight.detach(), emb0), "conditioner must be optimised too"
    assert not torch.equal(model._ema_flat, tr.P.flat) and int(model.model_ema.num_updates) == 3
    model.sync_trained_weights()
    eps1 = unet(x, t, context=ctx)
    assert not torch.equal(eps0, eps1), "the sampling program must see the trained weights"
    with model.ema_scope():
        eps_ema = unet(x, t, context=ctx)
    assert not torch.equal(eps_ema, eps1)


def test_log_images_state_dict_and_lightning_optimizer_follow_the_training_engine():
    """ImageLogger's hook and checkpoints during training (main.py:298-401, ddpm.py:1253-1361): after a step with a large
    learning rate, `log_images` must sample with the TRAINED EMA weights (they live in the engine's flat buffers), and
    `state_dict()` / `on_save_checkpoint` must hold the trained weights; the optimiser `configure_optimizers` hands to
    Lightning is the object `training_step` steps."""
    from helpers import make_fr_model
    model = make_fr_model(gain=0.25).train()
    model.cond_stage_trainable = True
    model.cond_stage_model.p_uncond = 0.0
    opt = model.configure_optimizers()
    assert opt is not None
    imgs = torch.tanh(rnd(70, 2, 128, 128, 3))
    batch = {"image": imgs, "class_label": torch.tensor([1, 5])}

    decoded = []
    real_decode = model.decode_first_stage
    model.decode_first_stage = lambda z, **kw: (decoded.append(z.detach().clone()), real_decode(z, **kw))[1]

    def logged():
        torch.manual_seed(1234)
        return model.log_images(batch, N=2, ddim_steps=4, ddim_eta=0.0)

    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    log0 = logged()
    assert set(log0) == {"inputs", "reconstruction", "samples"} and log0["samples"].shape == (2, 3, 128, 128)
    torch.testing.assert_close(logged()["samples"], log0["samples"], rtol=0, atol=0)       # same seed, same weights
    z = rnd(90, 2, 3, 32, 32).cuda()
    model.training_step_latents(z, {"class_label": batch["class_label"].cuda()}, lr=1e-3, t=torch.tensor([300, 800]).cuda(),
                                noise=rnd(93, 2, 3, 32, 32).cuda())
    assert model._cond_opt is opt and len(opt.state) > 0, "Lightning's optimiser must be the one that was stepped"
    log1 = logged()                                            # no explicit sync_trained_weights() call
    assert (log1["samples"] - log0["samples"]).abs().max().item() > 1e-3, "samples must reflect the training step"
    sd1 = model.state_dict()
    k_w, k_e = "model.diffusion_model.out.2.weight", "model_ema.diffusion_modelout2weight"
    assert not torch.equal(sd1[k_w], sd0[k_w]) and not torch.equal(sd1[k_e], sd0[k_e])
    tr = model.trainer()
    ema_sd = {k: v.cpu() for k, v in tr.state_dict_reference(model._ema_flat).items()}
    assert torch.equal(sd1[k_e].cpu(), ema_sd["out.2.weight"]) and torch.equal(sd1[k_w].cpu(), tr.state_dict_reference()["out.2.weight"].cpu())
    ckpt = {"state_dict": dict(sd0)}
    model.on_save_checkpoint(ckpt)
    assert torch.equal(ckpt["state_dict"][k_w], sd1[k_w]) and ckpt["ldmk_training_state"]["unet"]["step"] == 1
    # checker: the oracle's DDIM + decode on the synced EMA weights and the trained class embedding
    torch.manual_seed(1234)
    x_T = torch.randn((2, 3, 32, 32), device="cuda").cpu()
    c = model.cond_stage_model.embedding.weight.detach().cpu()[batch["class_label"]][:, None]
    lat = O.ddim_sample(ema_sd, W.FR_UNET, O.register_schedule(**W.SCHEDULE), 4, x_T, cond=c)
    # (compared before the quantiser: a latent within 1e-5 of a codebook cell boundary may legitimately decode to another code)
    torch.testing.assert_close(decoded[-1].cpu(), lat, rtol=5e-4, atol=5e-4)


def test_ema_shadow_resumes_from_the_checkpointed_model_ema_and_training_state_round_trips():
    """Resume semantics of LitEma (ema.py:25-44): the packed shadow starts from the `model_ema` buffers a checkpoint
    restored -- not from the live weights -- so one step gives decay*old_ema + (1-decay)*new_weights; and
    training_state()/load_training_state() carry the shadow and the conditioner optimiser."""
    from helpers import make_fr_model
    model = make_fr_model(gain=0.5).train()
    with torch.no_grad():                                 # a checkpoint whose EMA differs from the model
        for b_ in model.model_ema.buffers():
            if b_.dtype.is_floating_point and b_.dim() > 0:
                b_.mul_(0.75)
        model.model_ema.num_updates.fill_(1000)
    tr = model.trainer()
    pre = "diffusion_model."
    ema_sd = {k[len(pre):]: model.model_ema.shadow_of(k) for k in model.model_ema.m_name2s_name if k.startswith(pre)}
    back = tr.state_dict_reference(model._ema_flat)
    for k, v in ema_sd.items():
        assert torch.equal(back[k], v), k                 # pack_reference_state is the exact inverse
    old = model._ema_flat.clone()
    z = rnd(90, 2, 3, 32, 32).cuda()
    batch = {"class_label": torch.tensor([1, 5]).cuda()}
    model.cond_stage_model.p_uncond = 0.0
    model.training_step_latents(z, batch, lr=1e-5, t=torch.tensor([300, 800]).cuda(), noise=rnd(93, 2, 3, 32, 32).cuda())
    n_up = int(model.model_ema.num_updates)
    assert n_up == 1001
    decay = min(float(model.model_ema.decay), (1 + n_up) / (10 + n_up))
    want = old - (1.0 - decay) * (old - tr.P.flat)
    torch.testing.assert_close(model._ema_flat, want, rtol=1e-6, atol=1e-7)
    st = model.training_state()
    assert st["ema_flat"] is not None and st["_cond_opt"] is not None and st["unet"]["step"] == 1
    other = make_fr_model(gain=0.5).train()
    other.load_training_state(st)
    assert torch.equal(other._ema_flat, model._ema_flat) and torch.equal(other.trainer().P.flat, tr.P.flat)
    assert other._cond_opt is not None and other.trainer().P.step == 1


def test_p_losses_against_reference_fixture():
    """tests/golden/g9_p_losses.npz holds loss and gradients of the reference's own LatentDiffusion.p_losses +
    autograd (tools/make_golden.py --tree train).  The HIP step must reproduce them."""
    g = golden("g9_p_losses.npz")
    m, tr, sd, *_ = _setup(SMALL, 2, 16)
    x0, noise, ctx, t = rnd(101, 2, 3, 16, 16), rnd(102, 2, 3, 16, 16), rnd(103, 2, 1, 512), torch.tensor([17, 803])
    sched = O.register_schedule(**W.SCHEDULE)
    loss = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                       sched["sqrt_one_minus_alphas_cumprod"].cuda())
    assert abs(loss.item() - float(g["loss"])) <= 2e-5 * float(g["loss"])
    torch.testing.assert_close(tr.dctx.cpu().view(2, 1, 512), torch.from_numpy(g["dcontext"]), rtol=5e-4, atol=1e-7)
    direct = {"out.2.bias": tr.P.g["out.bpad"][:3]}
    for k in g.files:
        if k.startswith("grad:"):
            got = direct.get(k[5:], tr.P.g.get(k[5:]))
            ref = torch.from_numpy(g[k])
            torch.testing.assert_close(got.cpu(), ref, rtol=5e-4, atol=2e-6 * float(ref.abs().max()))
    # every parameter: L2 norm of the gradient (layout independent); fused buffers combine their parts
    stats = dict(zip([str(n) for n in g["names"]], g["stats"]))
    sq = lambda keys: sum(stats[k][1] ** 2 for k in keys) ** 0.5
    res = [p for p, mm in m._walk() if mm.kind == "res"]
    for name, gr in tr.P.g.items():
        if name == "emb_all":
            ref = sq([p + "emb_layers.1.weight" for p in res])
        elif name == "emb_all_b":
            ref = sq([p + "emb_layers.1.bias" for p in res])
        elif name.endswith(".qkv"):
            ref = sq([name[:-3] + f"attn1.to_{c}.weight" for c in "qkv"])
        elif name in ("te0", "te2"):
            ref = stats[f"time_embed.{name[2]}.weight"][1]
        elif name == "in.wpad":
            ref = stats["input_blocks.0.0.weight"][1]
        elif name == "out.wpad":
            ref = stats["out.2.weight"][1]
        elif name == "out.bpad":
            ref = stats["out.2.bias"][1]
        else:
            table = {"c1": "in_layers.2.weight", "c2": "out_layers.3.weight", "skip": "skip_connection.weight",
                     "pin": "proj_in.weight", "pout": "proj_out.weight", "o1": "attn1.to_out.0.weight",
                     "q2": "attn2.to_q.weight", "k2": "attn2.to_k.weight",
                     "v2": "attn2.to_v.weight", "o2": "attn2.to_out.0.weight", "ff2": "ff.net.2.weight",
                     "ff1n": "ff.net.0.proj.weight"}
            suf = name.rsplit(".", 1)[-1]
            if suf in table:
                ref = stats[name[:-len(suf)] + table[suf]][1]
            elif suf == "w":
                base = name[:-1]
                ref = stats[base + ("op.weight" if base + "op.weight" in stats else "conv.weight")][1]
            else:
                ref = stats[name][1]
        got = gr.double().norm().item()
        assert abs(got - ref) <= 3e-4 * ref + 1e-12, (name, got, ref)


def test_talking_face_unet_gradients_with_channel_concat():
    """TF model (ddpm2cond): 3 noisy-latent channels + 6 concat channels in, (B,1,1024) context; gradients of the
    reduced two-condition UNet against float64 autograd on the oracle."""
    cfg = dict(SMALL, in_channels=9, context_dim=1024)
    m, tr, sd, _, noise, ctx, t = _setup(cfg, 2, 16, seed=10)
    x0, c34 = rnd(21, 2, 3, 16, 16), rnd(22, 2, 6, 16, 16)
    sched = O.register_schedule(**W.SCHEDULE)
    sdg = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    a = sched["sqrt_alphas_cumprod"][t].view(-1, 1, 1, 1)
    b = sched["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    xin = torch.cat([a * x0 + b * noise, c34], 1).double()
    eps = O.unet_forward(sdg, cfg, xin, t, ctx.double())
    loss_ref = F.mse_loss(eps, noise.double())
    loss_ref.backward()
    loss = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                       sched["sqrt_one_minus_alphas_cumprod"].cuda(), c_concat=c34.cuda())
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * loss_ref.item()
    _check_all_grads(m, tr, {k: v.grad for k, v in sdg.items()}, 1e-4)


def _ddp_worker(rank, world, rdzv, q, overlap=False):
    import os
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    # gloo: both ranks share the one GPU of the box.  File rendezvous: a TCP port picked by the parent and released before
    # the children bind it can be taken in between (seen once in ~10 full-suite runs)
    dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)
    m, tr, sd, _, _, _, _ = _setup(SMALL, 2, 16)
    x0, noise, ctx = rnd(201, 4, 3, 16, 16), rnd(202, 4, 3, 16, 16), rnd(203, 4, 1, 512)
    t = torch.tensor([17, 803, 400, 999])
    lo, hi = 2 * rank, 2 * rank + 2                                   # contiguous shard of the global batch
    sched = O.register_schedule(**W.SCHEDULE)
    if overlap:      # buckets of >= 1 M floats handed to the all-reduce while the backward is still running
        x_noisy = None
        from dsml_thesis_amd import train_ops as T
        sa, sb = sched["sqrt_alphas_cumprod"].cuda(), sched["sqrt_one_minus_alphas_cumprod"].cuda()
        xn = T.q_sample(x0[lo:hi].cuda(), noise[lo:hi].cuda(), t[lo:hi].cuda(), sa, sb)
        tr.forward(xn, t[lo:hi].cuda(), ctx[lo:hi].cuda())
        tgt = torch.zeros_like(tr.eps_pad)
        tgt[..., :3] = noise[lo:hi].cuda().permute(0, 2, 3, 1)
        _, deps = T.mse_grad(tr.eps_pad, tgt, denom=noise[lo:hi].numel())
        tr.backward(deps, reduce_world=world, bucket_elems=1 << 20)
    else:
        tr.p_losses(x0[lo:hi].cuda(), ctx[lo:hi].cuda(), t[lo:hi].cuda(), noise[lo:hi].cuda(),
                    sched["sqrt_alphas_cumprod"].cuda(), sched["sqrt_one_minus_alphas_cumprod"].cuda())
        tr.all_reduce_grads(world)
    if rank == 0:
        q.put(tr.P.grad.cpu())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_data_parallel_gradients_equal_full_batch_gradients(overlap):
    """N1 multi-GPU contract: shard the batch over ranks, one all-reduce (mean) of the flat gradient buffer ==
    the gradient of the full batch on one rank (the loss is a mean over samples).  overlap=True: the bucketed
    reduction that runs inside backward() (tail buckets of the flat buffer are reduced while earlier layers compute)."""
    import os
    import tempfile
    import torch.multiprocessing as mp
    rdzv = os.path.join(tempfile.mkdtemp(prefix="ldmk_rdzv_"), "store")
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    procs = [ctx_mp.Process(target=_ddp_worker, args=(r, 2, rdzv, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    g2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, tr, sd, _, _, _, _ = _setup(SMALL, 2, 16)
    x0, noise, ctx = rnd(201, 4, 3, 16, 16), rnd(202, 4, 3, 16, 16), rnd(203, 4, 1, 512)
    t = torch.tensor([17, 803, 400, 999])
    sched = O.register_schedule(**W.SCHEDULE)
    tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                sched["sqrt_one_minus_alphas_cumprod"].cuda())
    g1 = tr.P.grad.cpu()
    err = (g1 - g2).abs().max().item() / g1.abs().max().item()
    assert err <= 2e-5, f"sharded-and-averaged vs full-batch gradients: {err:.3e}"


def test_differentiable_decode_first_stage_input_gradient():
    """N2: VQGAN decode with the straight-through quantiser; d(image)/d(latent) against float64 autograd on the oracle."""
    from dsml_thesis_amd.autoencoder import VQModelInterface
    from dsml_thesis_amd.train_decoder import DecoderGrad
    fs = W.VQ_F4
    vq = VQModelInterface(embed_dim=fs["embed_dim"], n_embed=fs["n_embed"], ddconfig=dict(fs["ddconfig"]),
                          lossconfig=dict(target="torch.nn.Identity"))
    vsd = W.synth_state_dict(W.vqmodel_param_shapes(fs))
    vq.load_state_dict(vsd, strict=False)
    vq = vq.cuda().eval()
    z = rnd(301, 1, 3, 8, 8)
    dec = DecoderGrad(vq)
    img = dec.forward(z.cuda())
    dimg = rnd(302, *img.shape)
    dz = dec.backward(dimg.cuda())
    v64 = {k: v.double() for k, v in vsd.items()}
    zz = z.double().requires_grad_(True)
    zq, _ = O.vq_quantize(zz.detach().float(), vsd["quantize.embedding.weight"])
    zst = zz + (zq.double() - zz).detach()                       # straight-through estimator, quantize.py:299
    ref = O.decoder_forward(v64, fs["ddconfig"], F.conv2d(zst, v64["post_quant_conv.weight"], v64["post_quant_conv.bias"]))
    ref.backward(dimg.double())
    torch.testing.assert_close(img.cpu().double(), ref.detach(), rtol=2e-4, atol=2e-4)
    err = (dz.cpu().double() - zz.grad).abs().max().item() / zz.grad.abs().max().item()
    assert err <= 1e-4, f"decoder input gradient: {err:.3e}"


def test_differentiable_ddim_two_steps_with_guidance():
    """N2: two eta=0 DDIM steps with classifier-free guidance (batch doubling) + differentiable decode; UNet parameter
    gradients accumulated over the steps and d(loss)/d(x_T) against float64 autograd on the oracle."""
    from dsml_thesis_amd.autoencoder import VQModelInterface
    from dsml_thesis_amd.schedule import ddim_step_table
    from dsml_thesis_amd.train_decoder import DecoderGrad, DifferentiableDDIM
    m, tr, sd, *_ = _setup(SMALL, 1, 8)
    fs = W.VQ_F4
    vq = VQModelInterface(embed_dim=fs["embed_dim"], n_embed=fs["n_embed"], ddconfig=dict(fs["ddconfig"]),
                          lossconfig=dict(target="torch.nn.Identity"))
    vsd = W.synth_state_dict(W.vqmodel_param_shapes(fs))
    vq.load_state_dict(vsd, strict=False)
    vq = vq.cuda().eval()
    sched = O.register_schedule(**W.SCHEDULE)
    ts = np.asarray([201, 601])
    table = ddim_step_table(sched["alphas_cumprod"], ts, 0.0)
    x_T, c, uc = rnd(311, 1, 3, 8, 8), rnd(312, 1, 1, 512), rnd(313, 1, 1, 512)
    scale = 2.0

    class _M:                                  # the two attributes DifferentiableDDIM reads from the LatentDiffusion
        scale_factor = 1.0
    dd = DifferentiableDDIM(_M(), trainer=tr, decoder=DecoderGrad(vq))
    img = dd.forward(x_T.cuda(), c.cuda(), table, ts, scale=scale, uc=uc.cuda())
    target = rnd(314, *img.shape)
    dimg = (2.0 / img.numel()) * (img - target.cuda())          # d mean((img-target)^2) / d img
    dx = dd.backward(dimg)

    sdg = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    v64 = {k: v.double() for k, v in vsd.items()}
    x = x_T.double().requires_grad_(True)
    xi = x
    for i in reversed(range(len(ts))):
        a_t, a_prev, sig, s1m = (float(v) for v in table[i])
        tt = torch.full((1,), int(ts[i]))
        e2 = O.unet_forward(sdg, SMALL, torch.cat([xi, xi]), torch.cat([tt, tt]), torch.cat([uc, c]).double())
        e_t = e2[:1] + scale * (e2[1:] - e2[:1])
        pred_x0 = (xi - s1m * e_t) / a_t ** 0.5                  # ddim2.py:281-289 with sigma = 0
        xi = a_prev ** 0.5 * pred_x0 + (1.0 - a_prev) ** 0.5 * e_t
    zq, _ = O.vq_quantize(xi.detach().float(), vsd["quantize.embedding.weight"])
    zst = xi + (zq.double() - xi).detach()
    ref = O.decoder_forward(v64, fs["ddconfig"], F.conv2d(zst, v64["post_quant_conv.weight"], v64["post_quant_conv.bias"]))
    loss = F.mse_loss(ref, target.double())
    loss.backward()
    torch.testing.assert_close(img.cpu().double(), ref.detach(), rtol=5e-4, atol=5e-4)
    err = (dx.cpu().double() - x.grad).abs().max().item() / x.grad.abs().max().item()
    assert err <= 5e-4, f"d loss / d x_T: {err:.3e}"
    worst = _check_all_grads(m, tr, {k: v.grad for k, v in sdg.items()}, 5e-4)
    print("worst accumulated gradient error", worst)


def test_latent_diffusion_clip_finetune_step():
    """LatentDiffusionCLIP surface (latent_diffclip.py:969-1033) with the l2 image loss: one fine-tune step through
    3 differentiable DDIM steps (strength 0.3, guidance 2) + decode lowers the loss on the same batch."""
    from helpers import fr_config
    from dsml_thesis_amd.util import instantiate_from_config
    from helpers import load_recipe
    cfg = fr_config(unet=SMALL)
    cfg.update(strength=0.3, num_train_steps=3, num_test_steps=4, unconditional_guidance_scale=2.0, cls_loss_w=0.0,
               clip_loss_w=0.0, id_loss_w=0.0, l2_loss_w=1.0, edit_attr="happy")
    model = instantiate_from_config({"target": "ldm.models.diffusion.latent_diffclip.LatentDiffusionCLIP", "params": cfg})
    load_recipe(model.model.diffusion_model)
    load_recipe(model.first_stage_model)
    load_recipe(model.cond_stage_model)
    model = model.cuda().train()
    assert list(model.train_ddim_timesteps) == [1, 150, 300] and model.trg == 1
    x = rnd(401, 1, 3, 16, 16).cuda()
    x0 = torch.tanh(rnd(402, 1, 3, 64, 64)).cuda()
    l0, ld = model.training_step_latents(x, ["face"], x0, lr=2e-6)
    assert set(ld) == {"train_l2_loss", "train_loss"} and torch.isfinite(l0)
    g = model.trainer().P.grad
    assert torch.isfinite(g).all() and g.abs().max().item() > 0
    l1, _ = model.training_step_latents(x, ["face"], x0, lr=2e-6)
    assert l1.item() < l0.item(), (l0.item(), l1.item())
    model.id_loss_w = 1.0
    with pytest.raises(NotImplementedError):
        model(x, ["face"], x0)


def test_differentiable_ddim_against_reference_fixture():
    """tests/golden/g10_diffclip.npz: the reference's ddim2.differentiable_p_sample_ddim x3 (guidance 2) +
    differentiable_decode_first_stage + l2 loss + autograd (tools/make_golden.py --tree diffclip)."""
    from dsml_thesis_amd.autoencoder import VQModelInterface
    from dsml_thesis_amd.schedule import ddim_step_table
    from dsml_thesis_amd.train_decoder import DecoderGrad, DifferentiableDDIM
    g = golden("g10_diffclip.npz")
    m, tr, sd, *_ = _setup(SMALL, 1, 16)
    fs = W.VQ_F4
    vq = VQModelInterface(embed_dim=fs["embed_dim"], n_embed=fs["n_embed"], ddconfig=dict(fs["ddconfig"]),
                          lossconfig=dict(target="torch.nn.Identity"))
    vq.load_state_dict(W.synth_state_dict(W.vqmodel_param_shapes(fs)), strict=False)
    vq = vq.cuda().eval()
    sched = O.register_schedule(**W.SCHEDULE)
    ts = g["timesteps"]
    table = ddim_step_table(sched["alphas_cumprod"], ts, 0.0)

    class _M:
        scale_factor = 1.0
    dd = DifferentiableDDIM(_M(), trainer=tr, decoder=DecoderGrad(vq))
    x, x0 = rnd(401, 1, 3, 16, 16).cuda(), torch.tanh(rnd(402, 1, 3, 64, 64)).cuda()
    img = dd.forward(x, rnd(403, 1, 1, 512).cuda(), table, ts, scale=2.0, uc=rnd(404, 1, 1, 512).cuda())
    torch.testing.assert_close(dd.z.cpu(), torch.from_numpy(g["z"]), rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(img.cpu(), torch.from_numpy(g["image"]).float(), rtol=2e-3, atol=2e-3)
    loss = F.mse_loss(img, x0)
    assert abs(loss.item() - float(g["loss"])) <= 5e-5 * float(g["loss"])
    dx = dd.backward((2.0 / img.numel()) * (img - x0))
    ref_dx = torch.from_numpy(g["dx"])
    assert (dx.cpu() - ref_dx).abs().max().item() <= 1e-3 * ref_dx.abs().max().item()
    stats = dict(zip([str(n) for n in g["names"]], g["stats"]))
    for name, ref in (("in.wpad", "input_blocks.0.0.weight"), ("te0", "time_embed.0.weight"),
                      ("output_blocks.1.0.c1", "output_blocks.1.0.in_layers.2.weight"),
                      ("middle_block.1.transformer_blocks.0.ff2", "middle_block.1.transformer_blocks.0.ff.net.2.weight"),
                      ("input_blocks.2.0.w", "input_blocks.2.0.op.weight"), ("out.0.weight", "out.0.weight")):
        got, want = tr.P.g[name].double().norm().item(), stats[ref][1]
        assert abs(got - want) <= 1e-3 * want, (name, got, want)


def test_optimizer_state_resume_is_bitwise():
    """two steps in one go == one step, save, restore into a fresh trainer, one more step."""
    sched = O.register_schedule(**W.SCHEDULE)
    sa, sb = sched["sqrt_alphas_cumprod"].cuda(), sched["sqrt_one_minus_alphas_cumprod"].cuda()

    def one(tr, x0, ctx, t, noise):
        tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sa, sb)
        tr.adamw_step(lr=1e-5)

    m, tr, sd, x0, noise, ctx, t = _setup(SMALL, 2, 16)
    one(tr, x0, ctx, t, noise)
    saved = tr.optimizer_state()
    one(tr, x0, ctx, t, noise)
    m2, tr2, *_ = _setup(SMALL, 2, 16)
    tr2.load_optimizer_state(saved)
    one(tr2, x0, ctx, t, noise)
    assert tr2.P.step == 2 and torch.equal(tr.P.flat, tr2.P.flat)
    assert torch.equal(tr.P.m, tr2.P.m) and torch.equal(tr.P.v, tr2.P.v)


def test_multi_token_context_gradients():
    """General cross-attention (context of 3 tokens: attn2.to_q / to_k / norm2 are live): every parameter gradient and
    d(loss)/d(context) against float64 autograd on the oracle."""
    m, tr, sd, x0, noise, _, t = _setup(SMALL, 2, 16, seed=20)
    ctx = rnd(29, 2, 3, 512)
    sched = O.register_schedule(**W.SCHEDULE)
    sdg = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    cg = ctx.double().requires_grad_(True)
    a = sched["sqrt_alphas_cumprod"][t].view(-1, 1, 1, 1)
    b = sched["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1, 1)
    eps = O.unet_forward(sdg, SMALL, (a * x0 + b * noise).double(), t, cg)
    loss_ref = F.mse_loss(eps, noise.double())
    loss_ref.backward()
    loss = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                       sched["sqrt_one_minus_alphas_cumprod"].cuda())
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * loss_ref.item()
    grads = {k: v.grad for k, v in sdg.items()}
    assert grads["input_blocks.1.1.transformer_blocks.0.attn2.to_q.weight"].abs().max().item() > 0
    _check_all_grads(m, tr, grads, 1e-4)
    err = (tr.dctx.double().cpu().view(2, 3, 512) - cg.grad).abs().max().item() / cg.grad.abs().max().item()
    assert err <= 1e-4, f"context gradient {err:.3e}"


def test_talking_face_training_step():
    """LatentDiffusion2Cond: class token + audio feature as one 1024-wide context token, masked-frame + identity latents
    on the channel axis; the step updates UNet, class embedder and EMA and returns the gradient of the audio feature."""
    from helpers import make_tf_model
    model = make_tf_model(gain=0.5).train()
    model.cond_stage_model_1.p_uncond = 0.0
    n = 2
    z, c34 = rnd(501, n, 3, 32, 32).cuda(), rnd(502, n, 6, 32, 32).cuda()
    audio = rnd(503, n, 1, 768).cuda()
    batch = {"class_label": torch.tensor([2, 6]).cuda()}
    emb0 = model.cond_stage_model_1.embedding.weight.detach().clone()
    losses = []
    for _ in range(3):
        loss, ld = model.training_step_latents(z, batch, audio, c34, lr=1e-5, t=torch.tensor([250, 750]).cuda(),
                                               noise=rnd(504, n, 3, 32, 32).cuda())
        losses.append(loss.item())
    assert losses[2] < losses[0] and ld["d_audio_feat"].shape == (n, 1, 768) and torch.isfinite(ld["d_audio_feat"]).all()
    assert ld["d_audio_feat"].abs().max().item() > 0
    assert not torch.equal(model.cond_stage_model_1.embedding.weight.detach(), emb0)
    # the same step with the raw 17-frame audio window: the window encoder is run and trained too
    win = rnd(505, n, 17, 768).cuda()
    w0 = model.cond_stage_model_2.attentionConvNet[0].weight.detach().clone()
    loss, ld = model.training_step_latents(z, batch, None, c34, lr=1e-5, t=torch.tensor([250, 750]).cuda(),
                                           noise=rnd(504, n, 3, 32, 32).cuda(), audio_window=win)
    assert torch.isfinite(loss) and not torch.equal(model.cond_stage_model_2.attentionConvNet[0].weight.detach(), w0)


def test_non_square_latent_gradients():
    """12x20 latent (240 / 60 tokens: ragged attention tiles, GroupNorm chunks that straddle the end of the image):
    every parameter gradient of the reduced UNet against float64 autograd on the oracle."""
    from dsml_thesis_amd.unet import UNetModel
    from dsml_thesis_amd.train import UNetTrainer
    m = UNetModel(**SMALL)
    sd = W.synth_state_dict(W.unet_param_shapes(SMALL))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    tr = UNetTrainer(m)
    x0, noise, ctx, t = rnd(601, 2, 3, 12, 20), rnd(602, 2, 3, 12, 20), rnd(603, 2, 1, 512), torch.tensor([5, 640])
    loss_ref, _, grads, dctx_ref, sched = _oracle_grads(SMALL, sd, x0, noise, ctx, t)
    loss = tr.p_losses(x0.cuda(), ctx.cuda(), t.cuda(), noise.cuda(), sched["sqrt_alphas_cumprod"].cuda(),
                       sched["sqrt_one_minus_alphas_cumprod"].cuda())
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    _check_all_grads(m, tr, grads, 1e-4)
