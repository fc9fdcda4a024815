"""Shared builders for the GPU tests / smoke / bench: native models loaded with the oracle's synthetic
weight recipe through the reference's state-dict keys."""
import torch

from oracle import weights as W


def fr_config(unet=None, vq=None):
    return dict(
        first_stage_config=dict(target="ldm.models.autoencoder.VQModelInterface",
                                params=dict(embed_dim=(vq or W.VQ_F4)["embed_dim"], n_embed=(vq or W.VQ_F4)["n_embed"],
                                            ddconfig=dict((vq or W.VQ_F4)["ddconfig"]),
                                            lossconfig=dict(target="torch.nn.Identity"))),
        cond_stage_config=dict(target="ldm.modules.encoders.modules.ClassEmbedder3",
                               params=dict(embed_dim=512, n_classes=8, key="class_label", p_uncond=0.2)),
        unet_config=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(unet or W.FR_UNET)),
        num_timesteps_cond=1, cond_stage_key="class_label", cond_stage_trainable=True, conditioning_key="crossattn",
        image_size=(unet or W.FR_UNET)["image_size"], channels=(unet or W.FR_UNET)["out_channels"],
        first_stage_key="image", log_every_t=200, monitor="val_loss_ema", **W.SCHEDULE)


def tf_config(seq_len=17):
    return dict(
        first_stage_config=fr_config()["first_stage_config"],
        cond_stage_config_1=dict(target="ldm.modules.encoders.modules.ClassEmbedder",
                                 params=dict(embed_dim=256, n_classes=8, key="class_label", p_uncond=0.2)),
        cond_stage_config_2=dict(target="ldm.modules.encoders.modules.Conv1DTemporalAttention",
                                 params=dict(seq_len=seq_len, subspace_dim=768, subspace2hidden=False)),
        unet_config=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(W.TF_UNET)),
        num_timesteps_cond=1, cond_stage_key_1="class_label", cond_stage_key_2="audio", cond_stage_trainable=True,
        conditioning_key="crossattn", image_size=32, channels=3, first_stage_key="image", log_every_t=200,
        monitor="val_loss_ema", **W.SCHEDULE)


def load_recipe(module, gain=1.0, seed=0):
    """Fill every floating-point tensor of `module` from the recipe, keyed by its own state-dict names."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point and v.dim() > 0}
    new = W.synth_state_dict(shapes, seed=seed, gain=gain)
    module.load_state_dict(new, strict=False)
    return new


def make_fr_model(gain=1.0, unet=None, vq=None, device="cuda"):
    from dsml_thesis_amd.ddpm import LatentDiffusion
    m = LatentDiffusion(**fr_config(unet, vq))
    load_recipe(m.model.diffusion_model, gain=gain)
    load_recipe(m.first_stage_model)
    load_recipe(m.cond_stage_model)
    return m.to(device).eval()


def make_tf_model(gain=1.0, seq_len=17, device="cuda"):
    from dsml_thesis_amd.ddpm import LatentDiffusion2Cond
    m = LatentDiffusion2Cond(**tf_config(seq_len))
    load_recipe(m.model.diffusion_model, gain=gain)
    load_recipe(m.first_stage_model)
    load_recipe(m.cond_stage_model_1)
    load_recipe(m.cond_stage_model_2)
    return m.to(device).eval()
