"""GPU: full UNetModel.forward (HIP program) against the reference's outputs (golden fixtures) and
the oracle, with the reference's state-dict keys loaded through load_state_dict."""
import numpy as np
import pytest
import torch

from conftest import golden, rnd
from oracle import ldm_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu


def make_unet(cfg, gain=1.0):
    from dsml_thesis_amd.unet import UNetModel
    m = UNetModel(**cfg)
    sd = W.synth_state_dict(W.unet_param_shapes(cfg), gain=gain)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.cuda().eval(), sd


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.float().cpu(), torch.as_tensor(np.asarray(b)).float(), rtol=rtol, atol=atol)


def test_resblock_golden_through_the_launch_program():
    """SURVEY row A7 on its own: ResBlock 160 -> 320 at 8x8 with the timestep embedding (openaimodel.py:255-275) against the
    reference's output (g3 `resblock`), built from the same NetBuilder calls UNetModel._build emits for a ResBlock:
    GroupNorm+SiLU pass, conv + per-sample emb vector + GroupNorm records from the epilogue, second GroupNorm+SiLU,
    1x1 skip projection, conv + residual."""
    from dsml_thesis_amd import ops
    from dsml_thesis_amd.engine import NetBuilder, Program
    g = golden("g3_ops.npz")
    keys = {}
    W._resblock(keys, "", 160, 320, 640)
    sd = {k: v.cuda() for k, v in W.synth_state_dict(keys, seed=3).items()}
    x, emb = rnd(14, 2, 160, 8, 8), rnd(15, 2, 640)
    n, h, w, hw = 2, 8, 8, 64
    pg = Program("cuda")
    nb = NetBuilder(pg, n)
    x0 = x.permute(0, 2, 3, 1).contiguous().cuda()
    emb_out = ops.dense_small(emb.cuda(), ops.pack_linear(sd["emb_layers.1.weight"]), sd["emb_layers.1.bias"], silu_in=True)
    c1, c2 = ops.pack_conv3x3(sd["in_layers.2.weight"]), ops.pack_conv3x3(sd["out_layers.3.weight"])
    skip_w = ops.pack_linear(sd["skip_connection.weight"])
    y1 = nb.gn_act(x0, None, hw, sd["in_layers.0.weight"], sd["in_layers.0.bias"], 1e-5)
    h1 = nb.conv(y1.view(n, h, w, 160), None, c1, sd["in_layers.2.bias"], h, w, batch_vec=emb_out.data_ptr(), bv_ld=320, stats=True)
    y2 = nb.gn_act(h1, None, hw, sd["out_layers.0.weight"], sd["out_layers.0.bias"], 1e-5).view(n, h, w, 320)
    skip = nb.lin(x0.view(n * hw, 160), skip_w, sd["skip_connection.bias"], hw)
    out = nb.conv(y2, None, c2, sd["out_layers.3.bias"], h, w, residual=skip, out=skip.view(n, h, w, 320), stats=True)
    pg.run()
    close(out.permute(0, 3, 1, 2), g["resblock"], 1e-4, 1e-4)
    first = out.clone()
    pg.run()                                   # the program is replayable: same buffers, same result
    assert torch.equal(out, first)


@pytest.mark.parametrize("pin", [None, (16, 2), (128, 2)])
@pytest.mark.parametrize("L_ctx", [1, 3])
def test_spatial_transformer_golden_through_the_launch_program(L_ctx, pin):
    """SURVEY row A8 on its own: SpatialTransformer(160, 5 heads x 32, context 512) at 8x8 (attention.py:218-261) against the
    reference's outputs (g3 `spatial_transformer`, 1 context token: the fast path; `spatial_transformer_L3`, 3 tokens: the
    general cross-attention), emitted by the same function UNetModel._build calls (unet.emit_spatial_transformer): GroupNorm
    folded into proj_in, LayerNorm folded into QKV / GEGLU, flash attention, the per-sample cross-attention vector.  `pin`:
    the job batch the plans are made for (None: this batch; 16 / 128: the plan sets of the benchmark / clip batches)."""
    from dsml_thesis_amd import unet as U
    from dsml_thesis_amd.engine import NetBuilder, Program
    g = golden("g3_ops.npz")
    keys = {}
    W._spatial_transformer(keys, "", 160, 5, 32, 1, 512)
    sd = {k: v.cuda() for k, v in W.synth_state_dict(keys, seed=7).items()}
    m = U._spatial_transformer(160, 5, 32, 1, 512)
    P = {}
    U.pack_spatial_transformer(P, sd, "", m)
    U.pack_gemm_copies(P)
    n, h, w = 2, 8, 8
    x, ctx = rnd(21, n, 160, h, w), rnd(22 if L_ctx == 1 else 23, n, L_ctx, 512)
    pg, ctx_pg = Program("cuda"), Program("cuda")
    nb = NetBuilder(pg, n, pin)
    x0 = x.permute(0, 2, 3, 1).contiguous().cuda()
    ctx_in = ctx.reshape(n * L_ctx, 512).cuda()
    out = U.emit_spatial_transformer(nb, ctx_pg, P, sd, "", m, x0, h, w, L_ctx, ctx_in, 512)
    names = [c[3] for c in pg.calls]
    assert ("ldmk_attn_cross" in names) == (L_ctx == 3) and sum(names.count(k) for k in ("ldmk_ln_stats_guard", "ldmk_ln_stats_ps", "ldmk_ln_stats_ps_h2")) == (2 if L_ctx == 1 else 3)
    ctx_pg.run()
    pg.run()
    close(out.permute(0, 3, 1, 2), g["spatial_transformer" if L_ctx == 1 else "spatial_transformer_L3"], 1e-4, 1e-4)
    first = out.clone()
    pg.run()
    assert torch.equal(out, first)


ROUTES = ["small", "batched"]


def _route(m, route, n, h, w, c_concat=0):
    """Pin the launch program the fixtures are held against: batch 1-2 takes the small-batch route (unet_small.py: slab
    GEMMs + ldmk_post, the reference's talking-face mode), a job batch of 16 the batched program (unet.py); "batched128" is
    the batched program with the plan set of the 128-frame clip (BASELINE configs[2]/[3]: policy_batch = 128)."""
    m.policy_batch = {"small": None, "batched": 16, "batched128": 128}[route]
    pg = m.program(n, h, w, 1, c_concat)
    names = [c[3] for c in pg.calls]
    if route == "small":
        assert getattr(pg, "small_route", False) and "ldmk_post" in names and "ldmk_attn_self_small" in names
        assert not {"ldmk_gn_finalize", "ldmk_gn_apply", "ldmk_ln_stats", "ldmk_gn_partial"} & set(names)
    else:
        assert not getattr(pg, "small_route", False) and "ldmk_gn_finalize" in names and "ldmk_post" not in names
    return pg


@pytest.mark.parametrize("route", ROUTES)
def test_unet_fr_golden(route):
    g = golden("g4_unet_fr.npz")
    m, _ = make_unet(W.FR_UNET)
    _route(m, route, 2, 32, 32)
    x, t, ctx = rnd(41, 2, 3, 32, 32), torch.tensor([3, 981]), rnd(42, 2, 1, 512)
    eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
    # fp32 end to end; different summation order than PyTorch-CPU over ~60 layers.  Measured (tools/parity_margin.py):
    # max |diff| 5.7e-6 on values up to 2.0 -- the bound below leaves a factor ~10 for other summation orders (plan changes)
    close(eps, g["fr_eps"], 3e-5, 3e-5)
    # replay of the same program is bitwise reproducible
    eps2 = m(x.cuda(), t.cuda(), context=ctx.cuda())
    assert torch.equal(eps, eps2)


@pytest.mark.parametrize("route", ROUTES + ["batched128"])
def test_unet_tf_concat_golden(route):
    g = golden("g7_talking_face.npz")
    m, _ = make_unet(W.TF_UNET)
    pg = _route(m, route, 2, 32, 32, 6)
    if route == "batched128":             # the clip's program: bf16x3 table shapes of the B = 128 rows, Winograd at every level
        from dsml_thesis_amd import lib as L
        gemms = [c[2] for c in pg.calls if c[3] == "ldmk_igemm"]
        assert sum(1 for a in gemms if a.compute in (L.COMPUTE_BF16X3, L.COMPUTE_F16X2)) >= 40
        names = [c[3] for c in pg.calls]
        # (Winograd, or -- the 32x32-level shapes of this batch -- the direct convolution on the conv-mode pre-split tile)
        assert sum(names.count(k) for k in ("ldmk_winograd_input", "ldmk_winograd_input_ps", "ldmk_winograd_input_ps_h2", "ldmk_gn_apply_ps_h2")) >= 20
    x, t = rnd(71, 2, 3, 32, 32), torch.tensor([11, 756])
    c12, c34 = rnd(72, 2, 1, 1024), rnd(73, 2, 6, 32, 32)
    eps = m(x.cuda(), t.cuda(), context=c12.cuda(), c_concat=c34.cuda())
    close(eps, g["tf_eps"], 3e-5, 3e-5)
    eps_cat = m(torch.cat([x, c34], 1).cuda(), t.cuda(), context=c12.cuda())
    assert torch.equal(eps, eps_cat)


@pytest.mark.parametrize("route", ROUTES)
def test_unet_northstar_64_golden(route):
    g = golden("g4_unet_fr.npz")
    m, _ = make_unet(W.NS_UNET)
    _route(m, route, 1, 64, 64)
    eps = m(rnd(43, 1, 4, 64, 64).cuda(), torch.tensor([501]).cuda(), context=rnd(44, 1, 1, 512).cuda())
    close(eps, g["ns_eps"], 3e-5, 3e-5)


def test_unet_winograd_route_against_the_reference_fixtures(monkeypatch):
    """The wide ResBlock convolutions take the Winograd F(2x2,3x3) route only from 256 tiles up (batch 16 at the
    benchmark sizes); the fixtures are batch 1-2.  Lower the threshold so that every eligible convolution of these small
    batches goes through it, and hold the result to the same reference fixtures and the same bound as the direct route."""
    from dsml_thesis_amd.engine import NetBuilder
    monkeypatch.setattr(NetBuilder, "WINO_MIN_TILES", 1)
    monkeypatch.setattr(NetBuilder, "UP_MIN_PIXELS", 1)
    monkeypatch.setenv("LDMK_PSC", "0")      # (round 5: no conv-mode pre-split tile here -- this test is about the Winograd route)
    g = golden("g4_unet_fr.npz")
    m, _ = make_unet(W.FR_UNET)
    m.policy_batch = 16                      # the batched program (batch 1-2 jobs take the small-batch route)
    x, t, ctx = rnd(41, 2, 3, 32, 32), torch.tensor([3, 981]), rnd(42, 2, 1, 512)
    eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
    launches = [c[3] for c in m.program(2, 32, 32, 1, 0).calls]
    assert sum(launches.count(k) for k in ("ldmk_winograd_input", "ldmk_winograd_input_ps", "ldmk_winograd_input_ps_h2")) >= 10 and "ldmk_gn_apply" in launches   # 160-channel convs stay direct
    assert sum(launches.count(k) for k in ("ldmk_upconv_gather", "ldmk_upconv_gather_ps", "ldmk_upconv_gather_ps_h2")) == 2      # both Upsample convolutions as four 2x2-tap phases
    close(eps, g["fr_eps"], 3e-5, 3e-5)
    assert torch.equal(eps, m(x.cuda(), t.cuda(), context=ctx.cuda()))
    m2, _ = make_unet(W.NS_UNET)
    m2.policy_batch = 16
    eps = m2(rnd(43, 1, 4, 64, 64).cuda(), torch.tensor([501]).cuda(), context=rnd(44, 1, 1, 512).cuda())
    close(eps, g["ns_eps"], 3e-5, 3e-5)


def test_unet_split_arithmetic_routes_against_the_reference_fixtures(monkeypatch):
    """The batched program runs its large GEMMs and its self attention in the fp32-accurate bf16x3 arithmetic (the shapes
    igemm_plans_x3.json lists).  (a) that route is what the golden tests above exercised; (b) with EVERY eligible GEMM forced
    into it (Winograd planes, upsample phases, GEGLU, LayerNorm-folded, GroupNorm prologues) the same fixtures hold with the
    same bound; (c) LDMK_SPLIT_BF16=0 gives the f32 matrix-core program, and the two agree far inside the bound."""
    from dsml_thesis_amd import engine, lib as L
    g = golden("g4_unet_fr.npz")
    x, t, ctx = rnd(41, 2, 3, 32, 32), torch.tensor([3, 981]), rnd(42, 2, 1, 512)

    def run(policy=16):
        m, _ = make_unet(W.FR_UNET)
        m.policy_batch = policy
        eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
        pg = m.program(2, 32, 32, 1, 0)
        n3 = sum(1 for c in pg.calls if c[3] == "ldmk_igemm" and c[2].compute in (L.COMPUTE_BF16X3, L.COMPUTE_F16X2))
        ng = sum(1 for c in pg.calls if c[3] == "ldmk_igemm")
        return eps, n3, ng, [c[3] for c in pg.calls]

    eps_a, n3, ng, names = run()
    assert n3 >= 40 and "ldmk_attn_self_x3" in names and {"ldmk_attn_self_h2", "ldmk_attn_self_h2_ps", "ldmk_attn_self_h2_tiles"} & set(names) and "ldmk_attn_self" not in names, (n3, ng)
    close(eps_a, g["fr_eps"], 3e-5, 3e-5)
    # (b) everything eligible
    monkeypatch.setattr(engine, "x3_plan", lambda a, m, far=False: (1, 1) if a.epi == L.EPI_GEGLU else (5, 1))
    eps_b, n3b, ngb, _ = run()
    assert n3b > n3 and n3b >= ngb - 4, (n3b, ngb)
    close(eps_b, g["fr_eps"], 3e-5, 3e-5)
    monkeypatch.undo()
    # (c) the f32 matrix-core program
    monkeypatch.setenv("LDMK_SPLIT_BF16", "0")
    engine.reset_tables()
    eps_c, n3c, _, names_c = run()
    assert n3c == 0 and "ldmk_attn_self" in names_c and not {"ldmk_attn_self_x3", "ldmk_attn_self_x3p", "ldmk_attn_self_h2", "ldmk_attn_self_h2_ps", "ldmk_attn_self_h2_tiles"} & set(names_c)
    close(eps_c, g["fr_eps"], 3e-5, 3e-5)
    assert (eps_a - eps_c).abs().max().item() < 1.5e-5
    monkeypatch.undo()
    engine.reset_tables()


def test_folded_layernorm_guard_switches_mean_dominated_models_to_the_unfolded_prologue():
    """LayerNorm is folded through the Linear behind it by default; on rows whose |mean| is many standard deviations that form
    cancels in fp32.  A checkpoint whose proj_in biases put the token rows at |mean| / std ~ 50 must still meet the oracle at the
    UNet bound with NO environment variable set: the statistics passes raise the guard flag, the first evaluation reads it,
    warns, re-packs with the unfolded prologue and evaluates again.  A well-conditioned model never leaves the folded form."""
    import warnings
    x, t, ctx = rnd(41, 2, 3, 32, 32), torch.tensor([3, 981]), rnd(42, 2, 1, 512)
    m, sd = make_unet(W.FR_UNET)
    m.policy_batch = 16                                  # the batched program (the small-batch route materialises LayerNorm anyway)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m(x.cuda(), t.cuda(), context=ctx.cuda())
    assert not m.ln_unfolded and m._ln_flag.item() == 0
    sd2 = {k: (v + 50.0 if k.endswith(".proj_in.bias") else v) for k, v in sd.items()}
    from dsml_thesis_amd.unet import UNetModel
    m2 = UNetModel(**W.FR_UNET)
    m2.load_state_dict(sd2, strict=True)
    m2 = m2.cuda().eval()
    m2.policy_batch = 16
    assert not m2.ln_unfolded
    with pytest.warns(RuntimeWarning, match="unfolded"):
        eps = m2(x.cuda(), t.cuda(), context=ctx.cuda())
    assert m2.ln_unfolded
    names = [c[3] for c in m2.program(2, 32, 32, 1, 0).calls]
    assert "ldmk_ln_stats_guard" not in names and names.count("ldmk_ln_stats") == 32
    ref = O.unet_forward(sd2, W.FR_UNET, x, t, ctx)
    close(eps, ref, 3e-5, 3e-5)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # decided once: later evaluations neither warn nor re-pack
        assert torch.equal(m2(x.cuda(), t.cuda(), context=ctx.cuda()), eps)


H2_ATTN = {"ldmk_attn_self_h2", "ldmk_attn_self_h2_ps", "ldmk_attn_self_h2_tiles"}


def _f16x2_launches(pg):
    """(launches that run in the F16X2 arithmetic, launches that run in a split arithmetic at all) of a launch program."""
    from dsml_thesis_amd import lib as L
    h2 = sum(1 for c in pg.calls if (c[3] == "ldmk_igemm" and c[2].compute == L.COMPUTE_F16X2) or c[3] in H2_ATTN)
    x3 = sum(1 for c in pg.calls if c[3] == "ldmk_igemm" and c[2].compute == L.COMPUTE_BF16X3)
    return h2, h2 + x3           # (the attention below ATTN_H2_MIN_TOKENS tokens is bf16x3 by plan: not counted)


def test_f16x2_range_flag_sends_one_site_back_to_bf16x3():
    """The F16X2 attention needs |K|, |V|, |scaled Q| < 1000.  A checkpoint whose attn1.to_k weights of ONE block are 3000 x larger
    (and its to_q weights as much smaller: same logits) must still meet the oracle with no environment variable set: the kernels
    raise that site's range flag (and saturate the operand, so nothing downstream overflows), the evaluation reads the flags,
    warns, re-plans THAT site -- LN1 -> QKV -> attention -> to_out of the block -- in bf16x3 and evaluates again; every other
    launch stays in F16X2.  A well-conditioned model stays in F16X2 throughout."""
    import warnings
    from dsml_thesis_amd.unet import UNetModel
    x, t, ctx = rnd(43, 2, 3, 32, 32), torch.tensor([5, 700]), rnd(44, 2, 1, 512)
    m, sd = make_unet(W.FR_UNET)
    m.policy_batch = 16
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m(x.cuda(), t.cuda(), context=ctx.cuda())
    st = m.arithmetic_status()
    assert st["f16x2"] and not st["denied"] and not st["flags_up"] and st["sites"] > 60
    pg0 = m.program(2, 32, 32, 1, 0)
    assert H2_ATTN & {c[3] for c in pg0.calls}
    h2_0, split_0 = _f16x2_launches(pg0)
    assert h2_0 == split_0 > 100               # every split-arithmetic launch of a healthy model runs in F16X2
    site = "input_blocks.1.1.transformer_blocks.0.attn1"
    key = site + ".to_k.weight"
    keyq = key.replace("to_k", "to_q")
    assert key in sd and keyq in sd
    # (K x 3000 with Q / 3000: the logits -- and the conditioning of the softmax -- are those of the original checkpoint)
    sd2 = {k: (v * 3000.0 if k == key else v / 3000.0 if k == keyq else v) for k, v in sd.items()}
    m2 = UNetModel(**W.FR_UNET)
    m2.load_state_dict(sd2, strict=True)
    m2 = m2.cuda().eval()
    m2.policy_batch = 16
    with pytest.warns(RuntimeWarning, match="F16X2"):
        eps = m2(x.cuda(), t.cuda(), context=ctx.cuda())
    st = m2.arithmetic_status()
    assert st["f16x2"] and st["denied"] == [site] and not st["flags_up"], st
    pg2 = m2.program(2, 32, 32, 1, 0)
    h2_2, split_2 = _f16x2_launches(pg2)
    assert split_0 - 3 <= split_2 <= split_0 and h2_0 - 4 <= h2_2 < h2_0, (h2_0, h2_2, split_0, split_2)      # the QKV, attention, to_out of the site: nothing else moved
    names0, names2 = [c[3] for c in pg0.calls], [c[3] for c in pg2.calls]
    assert sum(n in H2_ATTN for n in names2) == sum(n in H2_ATTN for n in names0) - 1
    close(eps, O.unet_forward(sd2, W.FR_UNET, x, t, ctx), 3e-5, 3e-5)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # decided once
        assert torch.equal(m2(x.cuda(), t.cuda(), context=ctx.cuda()), eps)
    # a model told up front (deny_f16x2) computes the same bits without ever raising a flag
    m3 = UNetModel(**W.FR_UNET)
    m3.load_state_dict(sd2, strict=True)
    m3 = m3.cuda().eval()
    m3.policy_batch = 16
    m3.deny_f16x2([site])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(m3(x.cuda(), t.cuda(), context=ctx.cuda()), eps)


def test_f16x2_outlier_residual_channel_moves_a_few_sites_only():
    """A checkpoint with ONE outlier channel (x 2000) in a transformer block's residual stream -- where real checkpoints keep such
    channels: proj_in's bias here -- through the batched program.  The LayerNorm-folded projections stage that stream RAW, so the
    sites that read it (LN1 -> QKV ..., LN3 -> GEGLU ...) leave the F16X2 range; because the staged value is saturated, not
    overflowed, nothing behind them sees a NaN and ONE repeat settles it: the oracle is met at the unchanged 3e-5, at most 10 % of
    the split-arithmetic launches changed arithmetic, and the step costs within 10 % of the all-F16X2 model's."""
    import time
    import warnings
    from dsml_thesis_amd.unet import UNetModel
    x, t, ctx = rnd(45, 2, 3, 32, 32), torch.tensor([3, 600]), rnd(46, 2, 1, 512)
    m, sd = make_unet(W.FR_UNET)
    m.policy_batch = 16
    sd2 = dict(sd)
    b = sd["input_blocks.2.1.proj_in.bias"].clone()
    b[7] = 2000.0
    sd2["input_blocks.2.1.proj_in.bias"] = b
    m2 = UNetModel(**W.FR_UNET)
    m2.load_state_dict(sd2, strict=True)
    m2 = m2.cuda().eval()
    m2.policy_batch = 16
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        eps = m2(x.cuda(), t.cuda(), context=ctx.cuda())
    h2w = [w for w in rec if "F16X2" in str(w.message)]
    assert 1 <= len(h2w) <= 2, [str(w.message) for w in rec]
    st = m2.arithmetic_status()
    assert st["f16x2"] and st["denied"] and all(d.startswith("input_blocks.2.1.") for d in st["denied"]), st
    close(eps, O.unet_forward(sd2, W.FR_UNET, x, t, ctx), 3e-5, 3e-5)
    h2_0, split_0 = _f16x2_launches(m.program(2, 32, 32, 1, 0))
    h2_2, split_2 = _f16x2_launches(m2.program(2, 32, 32, 1, 0))
    assert split_0 - 3 <= split_2 <= split_0 and h2_2 >= 0.9 * h2_0, (h2_0, h2_2, split_0, split_2)      # (a denied shape without a bf16x3 plan runs on the f32 cores)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(m2(x.cuda(), t.cuda(), context=ctx.cuda()), eps)

    def step_ms(model):
        pg = model.program(16, 32, 32, 1, 0)
        pg.ctx_program.run()
        for _ in range(3):
            pg.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pg.run()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 100.0
    t_ref, t_out = min(step_ms(m) for _ in range(3)), min(step_ms(m2) for _ in range(3))
    print(f"outlier-channel model: {t_out:.3f} ms per 16-sample step, all-F16X2 model {t_ref:.3f} ms; sites denied: {st['denied']}")
    assert t_out <= 1.10 * t_ref, (t_out, t_ref)


def test_f16x2_flag_is_read_on_every_forward_not_only_the_first():
    """The range flags are data dependent: the SECOND evaluation of a cached program, on an input a million times larger, drives the
    residual stream past 1000 where the stride-2 convolutions stage it raw.  forward() reads the flags on every call, re-plans those
    sites and returns finite values that meet the oracle -- not the saturated garbage of the first attempt."""
    import warnings
    x, t, ctx = rnd(47, 2, 3, 32, 32), torch.tensor([9, 400]), rnd(48, 2, 1, 512)
    m, sd = make_unet(W.FR_UNET)
    m.policy_batch = 16
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        e1 = m(x.cuda(), t.cuda(), context=ctx.cuda())
    close(e1, O.unet_forward(sd, W.FR_UNET, x, t, ctx), 3e-5, 3e-5)
    big = x * 1.0e6
    with pytest.warns(RuntimeWarning, match="F16X2"):
        e2 = m(big.cuda(), t.cuda(), context=ctx.cuda())
    st = m.arithmetic_status()
    assert st["f16x2"] and st["denied"] and not st["flags_up"], st
    ref = O.unet_forward(sd, W.FR_UNET, big, t, ctx)
    assert torch.isfinite(e2).all()
    close(e2, ref, 1e-4, 1e-4 * float(ref.abs().max()))
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # the small input again, on the re-planned program: still the oracle's
        close(m(x.cuda(), t.cuda(), context=ctx.cuda()), O.unet_forward(sd, W.FR_UNET, x, t, ctx), 3e-5, 3e-5)


def _student_t_state_dict(shapes, nu=3.0, seed=0):
    """The weight recipe of oracle/weights.py with Student-t (nu = 3) draws of the same variance in place of the Gaussian ones for
    every matrix / convolution weight: heavy tails -- a few weights per tensor 5-20 standard deviations out, as trained
    checkpoints have them.  Biases and norm parameters keep the recipe."""
    import zlib
    sd = W.synth_state_dict(shapes)
    for k, v in sd.items():
        if v.dim() >= 2:
            rs = np.random.RandomState((zlib.crc32(k.encode()) ^ (seed + 12345)) & 0xFFFFFFFF)
            tdraw = rs.standard_t(nu, size=tuple(v.shape)) / np.sqrt(nu / (nu - 2.0))
            sd[k] = torch.from_numpy((tdraw * float(v.std())).astype(np.float32))
    return sd


def test_f16x2_heavy_tailed_weights_below_the_range_threshold(monkeypatch):
    """Every parity margin of rounds 3-4 was measured on Gaussian weights.  Here the FR UNet carries Student-t (nu = 3) weights:
    outlier weights and, through them, outlier activation channels that stay BELOW the range threshold -- where the dropped lo lo
    term and the absolute-precision floor of F16X2 matter most.  eps must meet the oracle at the unchanged 3e-5 in F16X2, and the
    margin is logged next to the exact-split (bf16x3) model's."""
    import warnings
    from dsml_thesis_amd import engine
    from dsml_thesis_amd.unet import UNetModel
    x, t, ctx = rnd(49, 2, 3, 32, 32), torch.tensor([11, 850]), rnd(50, 2, 1, 512)
    sd = _student_t_state_dict(W.unet_param_shapes(W.FR_UNET))
    big = max(float(v.abs().max() / v.std()) for v in sd.values() if v.dim() >= 2)
    assert big > 20.0, big                                # the tails are there

    def run():
        m = UNetModel(**W.FR_UNET)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        m.policy_batch = 16
        with warnings.catch_warnings():
            warnings.simplefilter("error")               # below the threshold: no flag, no fall-back
            return m, m(x.cuda(), t.cuda(), context=ctx.cuda())
    ref = O.unet_forward(sd, W.FR_UNET, x, t, ctx)
    m, eps = run()
    st = m.arithmetic_status()
    assert st["f16x2"] and not st["denied"], st
    close(eps, ref, 3e-5, 3e-5)
    monkeypatch.setenv("LDMK_F16X2", "0")
    engine.reset_tables()
    try:
        m0, eps0 = run()
        assert not m0.f16x2
        close(eps0, ref, 3e-5, 3e-5)
    finally:
        monkeypatch.delenv("LDMK_F16X2")
        engine.reset_tables()
    d_h2, d_x3 = float((eps.cpu() - ref).abs().max()), float((eps0.cpu() - ref).abs().max())
    print(f"heavy-tailed weights (Student-t nu=3, max |w| / std = {big:.1f}): max |eps - reference| f16x2 {d_h2:.3e}, bf16x3 {d_x3:.3e} "
          f"(bound 3e-5 + 3e-5 |ref|, max |ref| {float(ref.abs().max()):.3f})")
    assert d_h2 <= 4.0 * max(d_x3, 2e-6)


@pytest.mark.parametrize("latent,batch", [(32, 3), (32, 5), (32, 7), (32, 24), (32, 48), (32, 96), (64, 2), (64, 4), (64, 6), (64, 48)])
def test_every_batch_gets_split_arithmetic_plans(latent, batch):
    """Plan coverage (round-4 review): tuned plans exist at B = 16 and B = 128; rounds 1-4 accepted a tuned row count only within 2 x,
    so B = 2-7 at 64x64x4 (an 8-way shard of a 32-sample job is B = 4) and 33-63 silently ran the f32 program (-44 %).  The nearest
    tuned plan is now carried whatever the distance: at every batch nearly all GEMM launches run in a split arithmetic, and eps of
    the first sample meets the oracle at the unchanged 3e-5 (a sample's result does not depend on its batch mates)."""
    from dsml_thesis_amd import lib as L
    cfg = W.FR_UNET if latent == 32 else W.NS_UNET
    m, sd = make_unet(cfg)
    ch = cfg["in_channels"]
    x1, t1, c1 = rnd(60 + batch, 1, ch, latent, latent), torch.tensor([321]), rnd(61 + batch, 1, 1, 512)
    x = torch.cat([x1, rnd(62, batch - 1, ch, latent, latent)])
    t = torch.cat([t1, torch.arange(batch - 1) * 37 % 1000])
    c = torch.cat([c1, rnd(63, batch - 1, 1, 512)])
    eps = m(x.cuda(), t.cuda(), context=c.cuda())
    pg = m.program(batch, latent, latent, 1, 0)
    comp = [a.compute for _, _, a, name in pg.calls if name == "ldmk_igemm"]
    split = sum(1 for f in comp if f in (L.COMPUTE_BF16X3, L.COMPUTE_F16X2))
    print(f"latent {latent} B = {batch}: {split} of {len(comp)} GEMM launches in a split arithmetic")
    from dsml_thesis_amd import unet_small
    if unet_small.wants_small_route(batch, latent, latent, 1):
        # up to 4096 token rows per job (B <= 4 at 32x32) the launch-bound program of DESIGN section 12 runs instead: slab GEMMs on
        # the f32 cores, chosen by measurement (profiles/r05_plan_coverage.txt has the A/B against the batched F16X2 program)
        assert "ldmk_attn_self_small" in {c[3] for c in pg.calls}
    else:
        # (a batch within 2x of a tuned one runs that batch's measured program, in which a few short-K projections stay on the f32
        #  row GEMM: 130 of 145 at B = 96 / 128; everywhere else the nearest plans are carried: 140 of 145)
        assert split >= 0.88 * len(comp), (split, len(comp))
    assert not m.arithmetic_status()["denied"]
    close(eps[:1], O.unet_forward(sd, cfg, x1, t1, c1), 3e-5, 3e-5)


@pytest.mark.parametrize("tag,cfg", [("h40", W.H40_UNET), ("h64", W.H64_UNET)])
def test_unet_attention_heads_that_are_not_32_wide(tag, cfg):
    """Reference kwargs `num_heads` / `num_head_channels` (openaimodel.py:443-469,542-549): a UNet with num_heads = 4 has heads 40 and
    80 wide, one with num_head_channels = 64 heads of 64.  Rounds 1-4 raised NotImplementedError; now the self attention of such
    heads runs as batched GEMMs on head-major, zero-padded copies (the flash kernels stay d = 32) and the multi-token cross
    attention on the width-templated kernel: eps against the REAL reference's outputs (g14), one- and three-token contexts."""
    g = golden("g14_variants.npz")
    m, sd = make_unet(cfg)
    assert not m._heads32
    x, t, ctx, ctx3 = rnd(150, 2, 3, 16, 16), torch.tensor([11, 870]), rnd(151, 2, 1, 512), rnd(152, 2, 3, 512)
    eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
    names = [c[3] for c in m.program(2, 16, 16, 1, 0).calls]
    assert "ldmk_heads_gather" in names and "ldmk_softmax_rows" in names and not any(n.startswith("ldmk_attn_self") for n in names)
    close(eps, g[tag + "_eps"], 3e-5, 3e-5)
    eps3 = m(x.cuda(), t.cuda(), context=ctx3.cuda())
    assert "ldmk_attn_cross_d" in [c[3] for c in m.program(2, 16, 16, 3, 0).calls]
    close(eps3, g[tag + "_eps_L3"], 3e-5, 3e-5)
    m.policy_batch = 16                      # the batched program's plans (split arithmetic on the GEMMs around the attention)
    close(m(x.cuda(), t.cuda(), context=ctx.cuda()), g[tag + "_eps"], 3e-5, 3e-5)


def test_class_conditional_unet_with_scale_shift_norm_and_new_attention_order():
    """Reference kwargs that raised NotImplementedError through round 4: `use_scale_shift_norm` (the timestep embedding modulates the
    second GroupNorm of every ResBlock, openaimodel.py:267-271 -- here a per-sample edit of that norm's coefficient planes,
    ldmk_gn_coef_film), `num_classes` (label embedding added to the timestep embedding, :513-514,726-728; DiffusionWrapper's 'adm'
    key, ddpm.py:1417-1420) and `use_new_attention_order` (QKVAttention, :379-407).  eps against the REAL reference's output (g14)."""
    from dsml_thesis_amd.ddpm import DiffusionWrapper
    g = golden("g14_variants.npz")
    m, sd = make_unet(W.ADM_UNET)
    x, t, y = rnd(153, 2, 3, 16, 16), torch.tensor([3, 512]), torch.tensor([7, 2])
    eps = m(x.cuda(), t.cuda(), y=y.cuda())
    names = [c[3] for c in m.program(2, 16, 16, 0, 0).calls]
    assert names.count("ldmk_gn_coef_film") == 8 and "ldmk_axpy" in names
    close(eps, g["adm_eps"], 3e-5, 3e-5)
    m.policy_batch = 16
    close(m(x.cuda(), t.cuda(), y=y.cuda()), g["adm_eps"], 3e-5, 3e-5)
    with pytest.raises(AssertionError, match="class-conditional"):
        m(x.cuda(), t.cuda())
    # through the wrapper: conditioning_key 'adm' hands the conditioning over as y
    w = DiffusionWrapper.__new__(DiffusionWrapper)
    torch.nn.Module.__init__(w)
    w.diffusion_model, w.conditioning_key = m, "adm"
    close(w(x.cuda(), t.cuda(), c_crossattn=[y.cuda()]), g["adm_eps"], 3e-5, 3e-5)


def test_attention_block_golden_through_the_launch_program():
    """AttentionBlock(160, 5 heads x 32) at 8x8 (openaimodel.py:278-324, QKVAttentionLegacy :347-372) against the reference's
    output (g13 `attention_block`), emitted by the function UNetModel._build calls for the unconditional UNet: GroupNorm folded
    into the qkv projection (output channels permuted from the reference's [head][q|k|v][32] order), flash attention with the
    logits scaled by d^-1/2 (the reference scales q and k by d^-1/4 each), proj_out + residual."""
    from dsml_thesis_amd import unet as U
    from dsml_thesis_amd.engine import NetBuilder, Program
    g = golden("g13_config0.npz")
    keys = {}
    W._attention_block(keys, "", 160)
    sd = {k: v.cuda() for k, v in W.synth_state_dict(keys, seed=11).items()}
    m = U._attention_block(160, 5)
    P = {}
    U.pack_attention_block(P, sd, "", m)
    U.pack_gemm_copies(P)
    n, h, w = 2, 8, 8
    x = rnd(131, n, 160, h, w)
    pg = Program("cuda")
    out = U.emit_attention_block(NetBuilder(pg, n), P, sd, "", m, x.permute(0, 2, 3, 1).contiguous().cuda(), h, w)
    pg.run()
    close(out.permute(0, 3, 1, 2), g["attention_block"], 1e-4, 1e-4)
    ref = O.attention_block({k: v.cpu() for k, v in sd.items()}, "", x, 5)
    close(out.permute(0, 3, 1, 2), ref, 1e-4, 1e-4)


@pytest.mark.parametrize("policy", [None, 16])
def test_unconditional_unet_golden(policy):
    """BASELINE configs[0] as worded: the UNet of a genuinely unconditional LDM (use_spatial_transformer=False, no context:
    AttentionBlock instead of SpatialTransformer) against the real reference's eps (g13 `uncond_eps`, B = 2 at 64x64x4), with
    the plans of this batch and with the plan set of the benchmark batch; conditional UNets still refuse context=None and this
    one refuses a context."""
    from dsml_thesis_amd import lib as L
    g = golden("g13_config0.npz")
    m, sd = make_unet(W.UNCOND_UNET, gain=0.25)
    m.policy_batch = policy
    x, t = rnd(130, 2, 4, 64, 64), torch.tensor([7, 640])
    eps = m(x.cuda(), t.cuda())
    close(eps, g["uncond_eps"], 3e-5, 3e-5)
    assert torch.equal(eps, m(x.cuda(), t.cuda(), context=None))
    names = [c[3] for c in m.program(2, 64, 64, 0, 0).calls]
    assert "ldmk_attn_cross" not in names and not any(n_.startswith("ldmk_ln_stats") for n_ in names)
    with pytest.raises(L.LdmkError, match="no cross-attention"):
        m(x.cuda(), t.cuda(), context=torch.zeros(2, 1, 512, device="cuda"))
    with pytest.raises(NotImplementedError):
        from dsml_thesis_amd.unet import UNetModel
        UNetModel(**dict(W.UNCOND_UNET, n_embed=8))


def test_unet_multi_token_context_vs_oracle():
    # L_ctx = 3 exercises the general cross-attention kernel (the shipped configs use L_ctx = 1)
    m, sd = make_unet(W.FR_UNET)
    x, t, ctx = rnd(45, 1, 3, 16, 16), torch.tensor([250]), rnd(46, 1, 3, 512)
    ref = O.unet_forward(sd, W.FR_UNET, x, t, ctx)
    close(m(x.cuda(), t.cuda(), context=ctx.cuda()), ref, 3e-5, 3e-5)


def test_unet_batch_invariance_and_ragged_batch():
    # per-sample results do not depend on what else is in the batch (the sharding argument)
    m, _ = make_unet(W.FR_UNET)
    x, t, ctx = rnd(47, 3, 3, 32, 32).cuda(), torch.tensor([5, 500, 995]).cuda(), rnd(48, 3, 1, 512).cuda()
    full = m(x, t, context=ctx)
    m.policy_batch = 3      # pin the GEMM tile shapes (hence the K-summation order) to the 3-sample job
    for i in range(3):
        one = m(x[i:i + 1], t[i:i + 1], context=ctx[i:i + 1])
        assert torch.equal(one[0], full[i])


def test_unet_rejects_unsupported():
    from dsml_thesis_amd.unet import UNetModel
    from dsml_thesis_amd import lib as L
    with pytest.raises(NotImplementedError):
        UNetModel(**dict(W.FR_UNET, dims=3))
    with pytest.raises(NotImplementedError):
        UNetModel(**dict(W.FR_UNET, n_embed=8))
    assert UNetModel(**dict(W.FR_UNET, resblock_updown=True)).resblock_updown      # (built since round 5: test_unet_resblock_updown)
    m, _ = make_unet(W.FR_UNET)
    with pytest.raises(L.LdmkError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 1, 512))


def test_small_route_launch_count():
    """The batch-1 step is a chain of dependent launches: the small-batch program must stay at <= 260 of them (the batched
    program issues ~415 kernels for the same evaluation) -- and <= 2 ldmk_post per GEMM-free stretch by construction."""
    m, _ = make_unet(W.FR_UNET)
    pg = _route(m, "small", 1, 32, 32)
    names = [c[3] for c in pg.calls]
    assert len(names) <= 260, len(names)
    assert names.count("ldmk_attn_self_small") == 16 and names.count("ldmk_post") <= 110


@pytest.mark.parametrize("policy", [None, 16])
@pytest.mark.parametrize("n,h,w", [(1, 24, 40), (3, 8, 8), (5, 16, 24)])
def test_unet_ragged_shapes_vs_oracle(n, h, w, policy):
    """Non-square / small latents and odd batch sizes: exercises partial tiles, the ragged attention tail (60 and
    4 tokens at the lowest level), the stand-alone GroupNorm statistics fallback (H*W not a multiple of 32) and
    split-K plans that differ from the benchmark shapes."""
    m, sd = make_unet(W.FR_UNET)
    m.policy_batch = policy                   # None: the small-batch route (all three jobs are below its row limit); 16: batched
    x, t, ctx = rnd(50, n, 3, h, w), torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(1)), rnd(51, n, 1, 512)
    ref = O.unet_forward(sd, W.FR_UNET, x, t, ctx)
    close(m(x.cuda(), t.cuda(), context=ctx.cuda()), ref, 1e-4, 1e-4)


def test_unet_empty_batch_fails_loudly():
    from dsml_thesis_amd import lib as L
    m, _ = make_unet(W.FR_UNET)
    with pytest.raises((L.LdmkError, RuntimeError, AssertionError)):
        m(torch.zeros(0, 3, 32, 32, device="cuda"), torch.zeros(0, dtype=torch.long, device="cuda"),
          context=torch.zeros(0, 1, 512, device="cuda"))


def test_unet_non_square_ragged_token_counts_vs_oracle():
    """24x40 latent: 960 / 240 / 60 tokens per level -- none a multiple of the 64-key attention tile, the two deeper levels
    not multiples of the 32-pixel GroupNorm chunk (stand-alone statistics pass, masked tiles everywhere)."""
    m, sd = make_unet(W.FR_UNET)
    x, t, ctx = rnd(47, 2, 3, 24, 40), torch.tensor([10, 990]), rnd(48, 2, 1, 512)
    eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
    ref = O.unet_forward(sd, W.FR_UNET, x, t, ctx)
    close(eps, ref, 1e-4, 1e-4)


def test_unet_resblock_updown():
    """Reference kwarg `resblock_updown` (openaimodel.py:570-584,660-674), the last UNetModel kwarg that raised NotImplementedError:
    ResBlock(down=True) / ResBlock(up=True) stand where Downsample / Upsample would; their parameter-free avg_pool2d(2, 2) /
    nearest x2 (:207-216,256-261) run as one ldmk_resample2 pass on SiLU(GroupNorm(x)) and one on the skip path.  eps against the
    REAL reference's outputs (g15): the shipped spatial-transformer UNet with one ResBlock per level at 32x32, and the
    class-conditional UNet with use_scale_shift_norm; with the batch's own plans and with the batched job's (policy_batch = 16)."""
    from dsml_thesis_amd import lib as L
    from dsml_thesis_amd import ops
    g = golden("g15_updown.npz")
    # the resampling passes on their own, against torch
    x = rnd(174, 2, 6, 10, 8).cuda()                         # NHWC [n][h][w][c], c % 4 == 0
    up, dn = torch.empty(2, 12, 20, 8, device="cuda"), torch.empty(2, 3, 5, 8, device="cuda")
    L.call("ldmk_resample2", x.data_ptr(), up.data_ptr(), 2, 6, 10, 8, 1, ops.stream())
    L.call("ldmk_resample2", x.data_ptr(), dn.data_ptr(), 2, 3, 5, 8, 0, ops.stream())
    nchw = x.permute(0, 3, 1, 2)
    assert torch.equal(up.permute(0, 3, 1, 2), torch.nn.functional.interpolate(nchw, scale_factor=2, mode="nearest"))
    close(dn.permute(0, 3, 1, 2), torch.nn.functional.avg_pool2d(nchw, 2, 2).cpu(), 1e-6, 1e-7)
    # the UNets
    m, sd = make_unet(W.UPDOWN_UNET)
    x, t, ctx = rnd(170, 2, 3, 32, 32), torch.tensor([5, 640]), rnd(171, 2, 1, 512)
    eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
    names = [c[3] for c in m.program(2, 32, 32, 1, 0).calls]
    assert names.count("ldmk_resample2") == 8                # 2 down + 2 up blocks, two passes each
    assert not any(k.endswith("op.weight") or k.endswith(".conv.weight") for k in m.state_dict())
    close(eps, g["ud_eps"], 3e-5, 3e-5)
    m.policy_batch = 16
    close(m(x.cuda(), t.cuda(), context=ctx.cuda()), g["ud_eps"], 3e-5, 3e-5)
    m, sd = make_unet(W.UPDOWN_ADM_UNET)
    x, t, y = rnd(173, 2, 3, 16, 16), torch.tensor([3, 512]), torch.tensor([7, 2])
    close(m(x.cuda(), t.cuda(), y=y.cuda()), g["ud_adm_eps"], 3e-5, 3e-5)
    m.policy_batch = 16
    close(m(x.cuda(), t.cuda(), y=y.cuda()), g["ud_adm_eps"], 3e-5, 3e-5)
    with pytest.raises(L.LdmkError, match="even sizes"):
        m(rnd(175, 2, 3, 15, 15).cuda(), t.cuda(), y=y.cuda())      # the down block would meet an odd grid: refused before any launch
