"""Container-only helper: make the reference's Python importable for fixture generation.

The reference needs third-party packages that are absent offline (omegaconf,
pytorch_lightning, torchvision, clip, kornia).  None of them executes on the sampling
path; this file registers empty stand-ins for those *third-party*
modules so that `import ldm...` succeeds.  No reference file is edited, copied or
stubbed.  Used only by tools/make_golden.py; never shipped to / imported on the GPU box.
"""
import importlib.machinery
import sys
import types

import torch.nn as nn


def _mod(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def install(tree):
    """tree: 'face_reenactment' | 'talking_face'."""
    sys.dont_write_bytecode = True
    if tree == "talking_face":
        import transformers  # noqa: F401  (must be imported before the torchvision stand-in)
    root = f"/root/reference/{tree}"
    if root not in sys.path:
        sys.path.insert(0, root)

    oc = _mod("omegaconf")
    lc = _mod("omegaconf.listconfig")

    class ListConfig(list):
        pass

    lc.ListConfig = ListConfig
    oc.listconfig = lc
    oc.ListConfig = ListConfig
    oc.OmegaConf = type("OmegaConf", (), {})      # name only: the driver scripts import it at module level
    _mod("albumentations")                        # dataset augmentation library, imported by taming/data/*.py

    pl = _mod("pytorch_lightning")

    class LightningModule(nn.Module):
        @property
        def device(self):
            for p in self.parameters():
                return p.device
            for b in self.buffers():
                return b.device
            import torch
            return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    plu = _mod("pytorch_lightning.utilities")
    plud = _mod("pytorch_lightning.utilities.distributed")
    plud.rank_zero_only = lambda f: f
    plu.distributed = plud
    pl.utilities = plu

    tv = _mod("torchvision")
    tvu = _mod("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    _mod("clip")
    _mod("kornia")
