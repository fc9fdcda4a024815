"""Synthetic-weight models for benchmarks, tools and smoke runs (no checkpoints exist offline).

Shipped hyper-parameters (values restated from face_reenactment/configs/latent-diffusion/affectnet-128-ldm-vq-f4.yaml:18-62
and talking_face/configs/latent-diffusion/mead-128-ldm-f4.yaml:19-63) and the SURVEY §8(c) weight recipe
`RandomState(crc32(key) ^ seed)`.  The oracle keeps its own copy of both (oracle/weights.py is test infrastructure and
is never imported from here); tests/test_host_logic.py checks that the two agree bit for bit."""
import zlib
from collections import OrderedDict

import numpy as np
import torch

FR_UNET = dict(image_size=32, in_channels=3, out_channels=3, model_channels=160, attention_resolutions=[4, 2, 1],
               num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=32, use_spatial_transformer=True,
               transformer_depth=1, context_dim=512)
TF_UNET = dict(FR_UNET, in_channels=9, context_dim=1024)
NS_UNET = dict(FR_UNET, image_size=64, in_channels=4, out_channels=4)          # north-star 64x64x4 latent
# BASELINE configs[0] as worded: a genuinely unconditional LDM -- no SpatialTransformer / context, AttentionBlock with 32-channel heads
UNCOND_UNET = dict(image_size=64, in_channels=4, out_channels=4, model_channels=160, attention_resolutions=[4, 2, 1],
                   num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=32)
VQ_F4 = dict(embed_dim=3, n_embed=16384,
             ddconfig=dict(double_z=False, z_channels=3, resolution=128, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4],
                           num_res_blocks=2, attn_resolutions=[32], dropout=0.0))
VQ_F4_256 = dict(embed_dim=4, n_embed=16384,
                 ddconfig=dict(double_z=False, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                               ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[64], dropout=0.0))
SCHEDULE = dict(timesteps=1000, linear_start=0.0015, linear_end=0.0205)


def synth_tensor(key, shape, seed=0, gain=1.0):
    """weights with >= 2 dims: N(0,1)/sqrt(fan_in)*gain (embedding tables: plain N(0,1)); biases: 0.02*N(0,1);
    1-D norm weights: 1 + 0.1*N(0,1).  MT19937 `RandomState` is bit-stable across NumPy versions and machines."""
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0xFFFFFFFF)
    n = rs.standard_normal(tuple(shape)).astype(np.float32)
    if len(shape) >= 2:
        if key.endswith("embedding.weight"):
            return n
        return (n * np.float32(gain / np.sqrt(int(np.prod(shape[1:]))))).astype(np.float32)
    if key.endswith(".bias"):
        return (n * np.float32(0.02)).astype(np.float32)
    return (np.float32(1.0) + np.float32(0.1) * n).astype(np.float32)


def synth_state_dict(shapes, seed=0, gain=1.0):
    return OrderedDict((k, torch.from_numpy(synth_tensor(k, s, seed=seed, gain=gain))) for k, s in shapes.items())


def load_recipe(module, gain=1.0, seed=0):
    """Fill every floating-point tensor of `module` from the recipe, keyed by its own state-dict names."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point and v.dim() > 0}
    new = synth_state_dict(shapes, seed=seed, gain=gain)
    module.load_state_dict(new, strict=False)
    return new


def fr_config(unet=None, vq=None):
    unet, vq = unet or FR_UNET, vq or VQ_F4
    return dict(
        first_stage_config=dict(target="ldm.models.autoencoder.VQModelInterface",
                                params=dict(embed_dim=vq["embed_dim"], n_embed=vq["n_embed"], ddconfig=dict(vq["ddconfig"]),
                                            lossconfig=dict(target="torch.nn.Identity"))),
        cond_stage_config=dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                               params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2)),
        unet_config=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(unet)),
        num_timesteps_cond=1, cond_stage_key="class_label", cond_stage_trainable=True, conditioning_key="crossattn",
        image_size=unet["image_size"], channels=unet["out_channels"], first_stage_key="image", log_every_t=200,
        monitor="val_loss_ema", **SCHEDULE)


def uncond_config(unet=None, vq=None):
    """cond_stage_config '__is_unconditional__' -> conditioning_key None (ddpm.py:443-444): DiffusionWrapper calls the UNet as
    diffusion_model(x, t)."""
    cfg = fr_config(unet or UNCOND_UNET, vq or VQ_F4_256)
    cfg.update(cond_stage_config="__is_unconditional__", cond_stage_trainable=False, conditioning_key=None)
    return cfg


def make_uncond_model(gain=1.0, unet=None, vq=None, device="cuda"):
    from .ddpm import LatentDiffusion
    m = LatentDiffusion(**uncond_config(unet, vq))
    load_recipe(m.model.diffusion_model, gain=gain)
    load_recipe(m.first_stage_model)
    return m.to(device).eval()


def tf_config(seq_len=17):
    return dict(
        first_stage_config=fr_config()["first_stage_config"],
        cond_stage_config_1=dict(target="ldm.modules.encoders.modules.ClassEmbedder",
                                 params=dict(embed_dim=256, n_classes=8, key="class_label", p_uncond=0.2)),
        cond_stage_config_2=dict(target="ldm.modules.encoders.modules.Conv1DTemporalAttention",
                                 params=dict(seq_len=seq_len, subspace_dim=768, subspace2hidden=False)),
        unet_config=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(TF_UNET)),
        num_timesteps_cond=1, cond_stage_key_1="class_label", cond_stage_key_2="audio", cond_stage_trainable=True,
        conditioning_key="crossattn", image_size=32, channels=3, first_stage_key="image", log_every_t=200,
        monitor="val_loss_ema", **SCHEDULE)


def make_fr_model(gain=1.0, unet=None, vq=None, device="cuda"):
    from .ddpm import LatentDiffusion
    m = LatentDiffusion(**fr_config(unet, vq))
    load_recipe(m.model.diffusion_model, gain=gain)
    load_recipe(m.first_stage_model)
    load_recipe(m.cond_stage_model)
    return m.to(device).eval()


def make_tf_model(gain=1.0, seq_len=17, device="cuda"):
    from .ddpm import LatentDiffusion2Cond
    m = LatentDiffusion2Cond(**tf_config(seq_len))
    load_recipe(m.model.diffusion_model, gain=gain)
    load_recipe(m.first_stage_model)
    load_recipe(m.cond_stage_model_1)
    load_recipe(m.cond_stage_model_2)
    return m.to(device).eval()
