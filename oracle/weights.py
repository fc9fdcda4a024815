"""TEST INFRASTRUCTURE ONLY (oracle/) -- synthetic seeded weights + state-dict key enumeration.

Nothing in the product package imports this file.  It is used by tests/, by
`__graft_entry__.smoke()` (the checker side) and by `bench.py`'s cpu_baseline leg.  The timed legs
build their synthetic models from dsml_thesis_amd/synth.py, which holds its own copy of the recipe and of
the shipped hyper-parameters (tests/test_host_logic.py checks that the two copies agree bit for bit).

Key enumeration follows the construction order of the reference modules:
  * UNetModel.__init__          /root/reference/face_reenactment/ldm/modules/diffusionmodules/openaimodel.py:443-692
  * SpatialTransformer.__init__ /root/reference/face_reenactment/ldm/modules/attention.py:226-248
  * Decoder/Encoder.__init__    /root/reference/face_reenactment/ldm/modules/diffusionmodules/model.py:368-533
  * VQModel.__init__            /root/reference/face_reenactment/ldm/models/autoencoder.py:15-63
The weight recipe is SURVEY.md §8(c): RandomState(crc32(key) ^ seed), applied to *all*
tensors including the ones the reference zero-initialises (otherwise a fresh UNet
outputs exact zeros and every comparison is vacuous).
"""
import zlib
from collections import OrderedDict

import numpy as np


# ----------------------------------------------------------------------------- key enumeration
def _resblock(keys, p, cin, cout, emb_ch, scale_shift=False):
    keys[p + "in_layers.0.weight"] = (cin,)
    keys[p + "in_layers.0.bias"] = (cin,)
    keys[p + "in_layers.2.weight"] = (cout, cin, 3, 3)
    keys[p + "in_layers.2.bias"] = (cout,)
    keys[p + "emb_layers.1.weight"] = ((2 if scale_shift else 1) * cout, emb_ch)      # use_scale_shift_norm: (scale | shift), openaimodel.py:218-224
    keys[p + "emb_layers.1.bias"] = ((2 if scale_shift else 1) * cout,)
    keys[p + "out_layers.0.weight"] = (cout,)
    keys[p + "out_layers.0.bias"] = (cout,)
    keys[p + "out_layers.3.weight"] = (cout, cout, 3, 3)
    keys[p + "out_layers.3.bias"] = (cout,)
    if cin != cout:
        keys[p + "skip_connection.weight"] = (cout, cin, 1, 1)
        keys[p + "skip_connection.bias"] = (cout,)


def _spatial_transformer(keys, p, ch, n_heads, d_head, depth, context_dim):
    inner = n_heads * d_head
    keys[p + "norm.weight"] = (ch,)
    keys[p + "norm.bias"] = (ch,)
    keys[p + "proj_in.weight"] = (inner, ch, 1, 1)
    keys[p + "proj_in.bias"] = (inner,)
    for d in range(depth):
        q = p + f"transformer_blocks.{d}."
        keys[q + "attn1.to_q.weight"] = (inner, inner)
        keys[q + "attn1.to_k.weight"] = (inner, inner)
        keys[q + "attn1.to_v.weight"] = (inner, inner)
        keys[q + "attn1.to_out.0.weight"] = (inner, inner)
        keys[q + "attn1.to_out.0.bias"] = (inner,)
        keys[q + "ff.net.0.proj.weight"] = (inner * 8, inner)
        keys[q + "ff.net.0.proj.bias"] = (inner * 8,)
        keys[q + "ff.net.2.weight"] = (inner, inner * 4)
        keys[q + "ff.net.2.bias"] = (inner,)
        cd = context_dim if context_dim is not None else inner
        keys[q + "attn2.to_q.weight"] = (inner, inner)
        keys[q + "attn2.to_k.weight"] = (inner, cd)
        keys[q + "attn2.to_v.weight"] = (inner, cd)
        keys[q + "attn2.to_out.0.weight"] = (inner, inner)
        keys[q + "attn2.to_out.0.bias"] = (inner,)
        for n in ("norm1", "norm2", "norm3"):
            keys[q + n + ".weight"] = (inner,)
            keys[q + n + ".bias"] = (inner,)
    keys[p + "proj_out.weight"] = (ch, inner, 1, 1)
    keys[p + "proj_out.bias"] = (ch,)


def _attention_block(keys, p, ch):
    """AttentionBlock (openaimodel.py:278-324): GroupNorm32, qkv = Conv1d(ch, 3 ch, 1), proj_out = Conv1d(ch, ch, 1)."""
    keys[p + "norm.weight"] = (ch,)
    keys[p + "norm.bias"] = (ch,)
    keys[p + "qkv.weight"] = (3 * ch, ch, 1)
    keys[p + "qkv.bias"] = (3 * ch,)
    keys[p + "proj_out.weight"] = (ch, ch, 1)
    keys[p + "proj_out.bias"] = (ch,)


def unet_layout(cfg):
    """Walk the UNet exactly like openaimodel.py:505-692 and return a block description.

    Returns dict(input=[...], middle=[...], output=[...]) where each entry is a list of
    layer tuples: ("conv", cin, cout) | ("res", cin, cout) | ("st", ch, heads, d_head) |
    ("down", ch) | ("up", ch) | ("attn", ch, heads, d_head) | ("res_down", ch, ch) | ("res_up", ch, ch) -- the last two are the
    ResBlock(down=True) / ResBlock(up=True) that stand where Downsample / Upsample would with resblock_updown=True
    (openaimodel.py:570-584,660-674).  The spatial-transformer configuration the shipped YAMLs use
    (use_spatial_transformer=True, num_head_channels set, legacy=True) and the unconditional variant
    (use_spatial_transformer=False: AttentionBlock / QKVAttentionLegacy, openaimodel.py:549-559) are covered.
    """
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    attn_res = set(cfg["attention_resolutions"])
    nhc = cfg.get("num_head_channels", -1)
    nh = cfg.get("num_heads", -1)

    st = cfg.get("use_spatial_transformer", False)
    kind = "st" if st else "attn"
    updown = bool(cfg.get("resblock_updown", False))

    def heads(ch):
        if nhc == -1:
            return nh, ch // nh
        n = ch // nhc
        return n, ch // n  # legacy branch, openaimodel.py:545-549 (AttentionBlock: heads = ch // num_head_channels, :297-301)

    inp = [[("conv", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                n, d = heads(ch)
                layers.append((kind, ch, n, d))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("res_down", ch, ch) if updown else ("down", ch)])
            chans.append(ch)
            ds *= 2
    n, d = heads(ch)
    mid = [("res", ch, ch), (kind, ch, n, d), ("res", ch, ch)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m)]
            ch = mc * m
            if ds in attn_res:
                n, d = heads(ch)
                layers.append((kind, ch, n, d))
            if level and i == nrb:
                layers.append(("res_up", ch, ch) if updown else ("up", ch))
                ds //= 2
            out.append(layers)
    return dict(input=inp, middle=mid, output=out, final_ch=ch)


def unet_param_shapes(cfg):
    mc = cfg["model_channels"]
    emb = 4 * mc
    depth = cfg.get("transformer_depth", 1)
    cd = cfg.get("context_dim")
    keys = OrderedDict()
    keys["time_embed.0.weight"] = (emb, mc)
    keys["time_embed.0.bias"] = (emb,)
    keys["time_embed.2.weight"] = (emb, emb)
    keys["time_embed.2.bias"] = (emb,)
    if cfg.get("num_classes") is not None:
        keys["label_emb.weight"] = (cfg["num_classes"], emb)       # openaimodel.py:513-514
    lay = unet_layout(cfg)

    def emit(prefix, layers):
        for j, l in enumerate(layers):
            p = f"{prefix}{j}."
            if l[0] == "conv":
                keys[p + "weight"] = (l[2], l[1], 3, 3)
                keys[p + "bias"] = (l[2],)
            elif l[0] in ("res", "res_down", "res_up"):     # (h_upd / x_upd of an up / down ResBlock carry no parameters)
                _resblock(keys, p, l[1], l[2], emb, bool(cfg.get("use_scale_shift_norm", False)))
            elif l[0] == "st":
                _spatial_transformer(keys, p, l[1], l[2], l[3], depth, cd)
            elif l[0] == "attn":
                _attention_block(keys, p, l[1])
            elif l[0] == "down":
                keys[p + "op.weight"] = (l[1], l[1], 3, 3)
                keys[p + "op.bias"] = (l[1],)
            elif l[0] == "up":
                keys[p + "conv.weight"] = (l[1], l[1], 3, 3)
                keys[p + "conv.bias"] = (l[1],)

    for i, layers in enumerate(lay["input"]):
        emit(f"input_blocks.{i}.", layers)
    emit("middle_block.", lay["middle"])
    for i, layers in enumerate(lay["output"]):
        emit(f"output_blocks.{i}.", layers)
    keys["out.0.weight"] = (lay["final_ch"],)
    keys["out.0.bias"] = (lay["final_ch"],)
    keys["out.2.weight"] = (cfg["out_channels"], mc, 3, 3)
    keys["out.2.bias"] = (cfg["out_channels"],)
    return keys


def _vq_resnet(keys, p, cin, cout):
    keys[p + "norm1.weight"] = (cin,)
    keys[p + "norm1.bias"] = (cin,)
    keys[p + "conv1.weight"] = (cout, cin, 3, 3)
    keys[p + "conv1.bias"] = (cout,)
    keys[p + "norm2.weight"] = (cout,)
    keys[p + "norm2.bias"] = (cout,)
    keys[p + "conv2.weight"] = (cout, cout, 3, 3)
    keys[p + "conv2.bias"] = (cout,)
    if cin != cout:
        keys[p + "nin_shortcut.weight"] = (cout, cin, 1, 1)
        keys[p + "nin_shortcut.bias"] = (cout,)


def _vq_attn(keys, p, c):
    keys[p + "norm.weight"] = (c,)
    keys[p + "norm.bias"] = (c,)
    for n in ("q", "k", "v", "proj_out"):
        keys[p + n + ".weight"] = (c, c, 1, 1)
        keys[p + n + ".bias"] = (c,)


def decoder_param_shapes(dd, prefix="decoder."):
    """model.py:462-533 (Decoder.__init__)."""
    ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(ch_mult)
    keys = OrderedDict()
    block_in = ch * ch_mult[-1]
    curr = dd["resolution"] // 2 ** (nres - 1)
    keys[prefix + "conv_in.weight"] = (block_in, dd["z_channels"], 3, 3)
    keys[prefix + "conv_in.bias"] = (block_in,)
    _vq_resnet(keys, prefix + "mid.block_1.", block_in, block_in)
    _vq_attn(keys, prefix + "mid.attn_1.", block_in)
    _vq_resnet(keys, prefix + "mid.block_2.", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * ch_mult[lvl]
        for ib in range(nrb + 1):
            _vq_resnet(keys, f"{prefix}up.{lvl}.block.{ib}.", block_in, block_out)
            block_in = block_out
            if curr in dd["attn_resolutions"]:
                _vq_attn(keys, f"{prefix}up.{lvl}.attn.{ib}.", block_in)
        if lvl != 0:
            keys[f"{prefix}up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            keys[f"{prefix}up.{lvl}.upsample.conv.bias"] = (block_in,)
            curr *= 2
    keys[prefix + "norm_out.weight"] = (block_in,)
    keys[prefix + "norm_out.bias"] = (block_in,)
    keys[prefix + "conv_out.weight"] = (dd["out_ch"], block_in, 3, 3)
    keys[prefix + "conv_out.bias"] = (dd["out_ch"],)
    return keys


def encoder_param_shapes(dd, prefix="encoder."):
    """model.py:368-432 (Encoder.__init__)."""
    ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(ch_mult)
    in_mult = (1,) + tuple(ch_mult)
    keys = OrderedDict()
    keys[prefix + "conv_in.weight"] = (ch, dd["in_channels"], 3, 3)
    keys[prefix + "conv_in.bias"] = (ch,)
    curr = dd["resolution"]
    block_in = ch
    for lvl in range(nres):
        block_in = ch * in_mult[lvl]
        block_out = ch * ch_mult[lvl]
        for ib in range(nrb):
            _vq_resnet(keys, f"{prefix}down.{lvl}.block.{ib}.", block_in, block_out)
            block_in = block_out
            if curr in dd["attn_resolutions"]:
                _vq_attn(keys, f"{prefix}down.{lvl}.attn.{ib}.", block_in)
        if lvl != nres - 1:
            keys[f"{prefix}down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            keys[f"{prefix}down.{lvl}.downsample.conv.bias"] = (block_in,)
            curr //= 2
    _vq_resnet(keys, prefix + "mid.block_1.", block_in, block_in)
    _vq_attn(keys, prefix + "mid.attn_1.", block_in)
    _vq_resnet(keys, prefix + "mid.block_2.", block_in, block_in)
    keys[prefix + "norm_out.weight"] = (block_in,)
    keys[prefix + "norm_out.bias"] = (block_in,)
    zc = 2 * dd["z_channels"] if dd.get("double_z", True) else dd["z_channels"]
    keys[prefix + "conv_out.weight"] = (zc, block_in, 3, 3)
    keys[prefix + "conv_out.bias"] = (zc,)
    return keys


def vqmodel_param_shapes(fs):
    """autoencoder.py:15-63 (VQModel.__init__), key order as registered."""
    dd = fs["ddconfig"]
    keys = OrderedDict()
    keys.update(encoder_param_shapes(dd))
    keys.update(decoder_param_shapes(dd))
    keys["quantize.embedding.weight"] = (fs["n_embed"], fs["embed_dim"])
    keys["quant_conv.weight"] = (fs["embed_dim"], dd["z_channels"], 1, 1)
    keys["quant_conv.bias"] = (fs["embed_dim"],)
    keys["post_quant_conv.weight"] = (dd["z_channels"], fs["embed_dim"], 1, 1)
    keys["post_quant_conv.bias"] = (dd["z_channels"],)
    return keys


def audio_attention_param_shapes(seq_len, subspace_dim=768, prefix=""):
    """talking_face/ldm/modules/encoders/modules.py:76-101 (Conv1DTemporalAttention)."""
    keys = OrderedDict()
    chans = [subspace_dim, 192, 64, 16, 4, 1]
    for i in range(5):
        keys[f"{prefix}attentionConvNet.{2 * i}.weight"] = (chans[i + 1], chans[i], 3)
        keys[f"{prefix}attentionConvNet.{2 * i}.bias"] = (chans[i + 1],)
    keys[prefix + "attentionNet.0.weight"] = (seq_len, seq_len)
    keys[prefix + "attentionNet.0.bias"] = (seq_len,)
    return keys


# ----------------------------------------------------------------------------- weight recipe
def synth_tensor(key, shape, seed=0, gain=1.0):
    """One tensor of the SURVEY §8(c) recipe (numpy float32).

    weights with >=2 dims: N(0,1)/sqrt(fan_in) * gain; biases: N(0,1)*0.02;
    norm weights (1-D '.weight'): 1 + 0.1*N(0,1).  `RandomState` (MT19937) is bit-stable
    across NumPy versions, so the GPU box regenerates identical values.
    """
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0xFFFFFFFF)
    n = rs.standard_normal(tuple(shape)).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if key.endswith("embedding.weight"):
            return n  # nn.Embedding tables: plain N(0,1) like torch's default
        return (n * np.float32(gain / np.sqrt(fan_in))).astype(np.float32)
    if key.endswith(".bias"):
        return (n * np.float32(0.02)).astype(np.float32)
    return (np.float32(1.0) + np.float32(0.1) * n).astype(np.float32)


def synth_state_dict(shapes, seed=0, gain=1.0, as_torch=True):
    out = OrderedDict()
    for k, s in shapes.items():
        a = synth_tensor(k, s, seed=seed, gain=gain)
        if as_torch:
            import torch

            a = torch.from_numpy(a)
        out[k] = a
    return out


# ----------------------------------------------------------------------------- shipped configs
# Hyper-parameters restated from the reference YAMLs (values only):
#   face_reenactment/configs/latent-diffusion/affectnet-128-ldm-vq-f4.yaml:18-40,42-62
#   talking_face/configs/latent-diffusion/mead-128-ldm-f4.yaml:19-41,43-63
FR_UNET = dict(image_size=32, in_channels=3, out_channels=3, model_channels=160,
               attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4],
               num_head_channels=32, use_spatial_transformer=True, transformer_depth=1,
               context_dim=512)
TF_UNET = dict(FR_UNET, in_channels=9, context_dim=1024)
# north-star variant: same code, 64x64x4 latent
NS_UNET = dict(FR_UNET, image_size=64, in_channels=4, out_channels=4)
# attention head widths other than 32 (reference kwargs num_heads / num_head_channels, openaimodel.py:443-469,542-549): small UNets
# for the g14 fixtures.  num_heads = 4: legacy dim_head = ch // num_heads = 40 at 160 channels, 80 at 320; num_head_channels = 64:
# 2 / 4 heads of 64 at 128 / 256 channels
H40_UNET = dict(image_size=16, in_channels=3, out_channels=3, model_channels=160, attention_resolutions=[1, 2], num_res_blocks=1,
                channel_mult=[1, 2], num_heads=4, use_spatial_transformer=True, transformer_depth=1, context_dim=512)
# a class-conditional ('adm') UNet with the other reference kwargs no shipped YAML sets: use_scale_shift_norm (FiLM through the second
# GroupNorm), num_classes (label embedding added to the timestep embedding), use_new_attention_order (QKVAttention)
ADM_UNET = dict(image_size=16, in_channels=3, out_channels=3, model_channels=64, attention_resolutions=[1, 2], num_res_blocks=1,
                channel_mult=[1, 2], num_head_channels=32, use_scale_shift_norm=True, num_classes=10, use_new_attention_order=True)
H64_UNET = dict(image_size=16, in_channels=3, out_channels=3, model_channels=128, attention_resolutions=[1, 2], num_res_blocks=1,
                channel_mult=[1, 2], num_head_channels=64, use_spatial_transformer=True, transformer_depth=1, context_dim=512)
# resblock_updown (openaimodel.py:570-584,660-674): ResBlock(down=True) / ResBlock(up=True) instead of Downsample / Upsample -- on the
# shipped spatial-transformer UNet with one ResBlock per level (two of each at 32x32) and, together with use_scale_shift_norm, on
# the class-conditional one
UPDOWN_UNET = dict(FR_UNET, num_res_blocks=1, resblock_updown=True)
UPDOWN_ADM_UNET = dict(ADM_UNET, resblock_updown=True)
# BASELINE configs[0] as worded: a genuinely UNCONDITIONAL LDM (cond_stage_config "__is_unconditional__" -> conditioning_key None,
# ddpm.py:443-444): no SpatialTransformer, AttentionBlock / QKVAttentionLegacy with 32-channel heads, no context
UNCOND_UNET = dict(image_size=64, in_channels=4, out_channels=4, model_channels=160, attention_resolutions=[4, 2, 1],
                   num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=32)
VQ_F4 = dict(embed_dim=3, n_embed=16384,
             ddconfig=dict(double_z=False, z_channels=3, resolution=128, in_channels=3, out_ch=3,
                           ch=128, ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[32],
                           dropout=0.0))
VQ_F4_256 = dict(embed_dim=4, n_embed=16384,
                 ddconfig=dict(double_z=False, z_channels=4, resolution=256, in_channels=3, out_ch=3,
                               ch=128, ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[64],
                               dropout=0.0))
SCHEDULE = dict(timesteps=1000, linear_start=0.0015, linear_end=0.0205)
