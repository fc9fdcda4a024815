"""Config factory: the reference's plugin boundary (`ldm/util.py:78-93`).  A YAML `target:` string naming
a reference class (e.g. `ldm.modules.diffusionmodules.openaimodel.UNetModel`) is resolved to the MI355X-native
class of this package, so shipped configs instantiate unchanged; `dsml_thesis_amd.*` targets work directly."""
import importlib

# reference dotted path -> native class path
TARGET_MAP = {
    "ldm.modules.diffusionmodules.openaimodel.UNetModel": "dsml_thesis_amd.unet.UNetModel",
    "ldm.models.autoencoder.VQModelInterface": "dsml_thesis_amd.autoencoder.VQModelInterface",
    "ldm.models.diffusion.ddpm.LatentDiffusion": "dsml_thesis_amd.ddpm.LatentDiffusion",
    "ldm.models.diffusion.ddpm2cond.LatentDiffusion": "dsml_thesis_amd.ddpm.LatentDiffusion2Cond",
    "ldm.models.diffusion.ddpm.DiffusionWrapper": "dsml_thesis_amd.ddpm.DiffusionWrapper",
    "ldm.models.diffusion.latent_diffclip.LatentDiffusionCLIP": "dsml_thesis_amd.latent_diffclip.LatentDiffusionCLIP",
    "ldm.modules.encoders.modules.ClassEmbedder3": "dsml_thesis_amd.encoders.ClassEmbedder3",
    "ldm.modules.encoders.modules.ClassEmbedder": "dsml_thesis_amd.encoders.ClassEmbedder",
    "ldm.modules.encoders.modules.Conv1DTemporalAttention": "dsml_thesis_amd.encoders.Conv1DTemporalAttention",
    "ldm.models.diffusion.ddim.DDIMSampler": "dsml_thesis_amd.ddim.DDIMSampler",
    "ldm.models.diffusion.ddim2cond.DDIMSampler": "dsml_thesis_amd.ddim.DDIMSampler",
}


def get_obj_from_str(string, reload=False):
    string = TARGET_MAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def load_yaml_config(path):
    """Plain-dict loader for the reference YAMLs (OmegaConf is optional and absent offline)."""
    import yaml
    with open(path) as fh:
        return yaml.safe_load(fh)
