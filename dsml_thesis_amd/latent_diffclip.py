"""`LatentDiffusionCLIP` (face_reenactment/ldm/models/diffusion/latent_diffclip.py:430-1003): fine-tune the UNet through
a few differentiable DDIM steps + the differentiable first stage, with image-space losses.

The UNet / decoder forward and backward run on libldmk kernels (`train.py`, `train_decoder.py`).  The image-space
losses are PyTorch modules working on the decoded image, exactly where the reference puts them: l2 is built in;
the ArcFace identity loss, the CLIP directional loss and the emotion-classifier loss need pretrained networks that are
not part of this path -- they are plain callables the caller plugs in (`id_loss_func(x, x0)`, `clip_loss_func(x0,
src_label, x, trg_txt)`, `cls_loss_func(x, trg)`); a non-zero weight without its callable raises."""
import numpy as np
import torch
import torch.nn as nn

from .ddpm import LatentDiffusion
from .schedule import ddim_step_table, make_ddim_timesteps_strength
from .train_decoder import DifferentiableDDIM

EMOTIONS = {"neutral": 0, "happy": 1, "sad": 2, "surprised": 3, "scared": 4, "disgusted": 5, "angry": 6}


class LatentDiffusionCLIP(LatentDiffusion):
    def __init__(self, first_stage_config, cond_stage_config, strength=0.5, num_train_steps=6, num_test_steps=40, eta=0.0,
                 temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True,
                 quantize_x0=False, unconditional_guidance_scale=1., cls_loss_w=1.0, clip_loss_w=1.0, id_loss_w=1.0,
                 l2_loss_w=1.0, cls_model_name="enet_b2_7", clip_model_name="ViT-B/16", edit_attr=None, **kwargs):
        assert edit_attr is not None
        if eta != 0.0 or quantize_x0 or score_corrector is not None:
            raise NotImplementedError("LatentDiffusionCLIP: eta = 0, no x0 quantisation, no score corrector (shipped settings)")
        super().__init__(first_stage_config, cond_stage_config, **kwargs)
        self.strength, self.num_train_steps, self.num_test_steps = strength, num_train_steps, num_test_steps
        self.unconditional_guidance_scale = unconditional_guidance_scale
        self.cls_loss_w, self.clip_loss_w, self.id_loss_w, self.l2_loss_w = cls_loss_w, clip_loss_w, id_loss_w, l2_loss_w
        self.l2_loss_func = nn.MSELoss()
        self.id_loss_func = self.clip_loss_func = self.cls_loss_func = None        # plugged in by the caller
        self.edit_attr = edit_attr
        self.trg_txts = [f"{edit_attr} face"]                                       # utils/text_dic.py: ('face', '<emotion> face')
        self.trg = EMOTIONS[edit_attr]                                              # latent_diffclip.py:541-544
        self.label2emotion_dict = {0: "face", 1: "happy face", 2: "sad face", 3: "surprised face", 4: "scared face",
                                   5: "disgusted face", 6: "angry face", 7: "face"}
        self.label2emotion_dict[self.trg] = "face"
        self.train_ddim_timesteps = make_ddim_timesteps_strength(num_train_steps, self.num_timesteps, strength)
        self.test_ddim_timesteps = make_ddim_timesteps_strength(num_test_steps, self.num_timesteps, strength)
        self._ddd = None
        self._table_cache = {}

    def _tables(self, training):
        """(timesteps, [S][4] coefficient rows), built on the host once per mode (no device sync in the step)."""
        key = bool(training)
        if key not in self._table_cache:
            ts = self.train_ddim_timesteps if training else self.test_ddim_timesteps
            self._table_cache[key] = (np.asarray(ts), ddim_step_table(self.alphas_cumprod.detach().cpu(), ts, 0.0))
        return self._table_cache[key]

    def differentiable(self):
        if self._ddd is None:
            self._ddd = DifferentiableDDIM(self)
        return self._ddd

    def clip_losses(self, x, x0, src_label):
        """latent_diffclip.py:1005-1033 on the decoded image x (requires_grad leaf) and the original x0."""
        prefix = "train" if self.training else "val"
        loss_dict = {}
        zero = x.new_zeros(())
        l2 = self.l2_loss_func(x, x0) if self.l2_loss_w else zero
        loss_dict[f"{prefix}_l2_loss"] = l2

        def need(func, name):
            if func is None:
                raise NotImplementedError(f"LatentDiffusionCLIP: {name}_loss_w != 0 needs a `{name}_loss_func` callable "
                                          f"(the pretrained network is not part of this package)")
            return func
        idl = need(self.id_loss_func, "id")(x, x0) if self.id_loss_w else zero
        if self.id_loss_w:
            loss_dict[f"{prefix}_id_loss"] = idl
        cl = zero
        if self.clip_loss_w:
            cl = -torch.log((2 - need(self.clip_loss_func, "clip")(x0, src_label, x, self.trg_txts[0])) / 2)
            loss_dict[f"{prefix}_clip_loss"] = cl
        cls = need(self.cls_loss_func, "cls")(x, self.trg) if self.cls_loss_w else zero
        if self.cls_loss_w:
            loss_dict[f"{prefix}_cls_loss"] = cls
        loss = self.l2_loss_w * l2 + self.id_loss_w * idl + self.clip_loss_w * cl + self.cls_loss_w * cls
        loss_dict[f"{prefix}_loss"] = loss
        return loss, loss_dict

    def forward(self, x, src_label, x0, *args, **kwargs):
        """latent_diffclip.py:969-1003: x = (inverted) latents, x0 = original images.  Runs the differentiable DDIM +
        decode on the HIP kernels, the losses under torch autograd on the decoded image, then the hand-written backward:
        UNet parameter gradients are left in `self.trainer().P.grad`.  Returns (loss, loss_dict)."""
        assert x0 is not None
        b, dev = x.shape[0], x.device
        scale = self.unconditional_guidance_scale
        uc = None
        if scale > 1.0:
            uc = self.cond_stage_model.uncond_embedding(torch.zeros(b, 1, dtype=torch.long, device=dev)).detach()
        c_trg = self.cond_stage_model.embedding(torch.full((b, 1), self.trg, dtype=torch.long, device=dev)).detach()
        ts, table = self._tables(self.training)
        dd = self.differentiable()
        img = dd.forward(x, c_trg, table, np.asarray(ts), scale=scale, uc=uc)
        with torch.enable_grad():
            leaf = img.detach().requires_grad_(True)
            loss, loss_dict = self.clip_losses(leaf, x0.to(dev).float(), src_label)
            loss.backward()
        dd.backward(leaf.grad)
        return loss.detach(), {k: v.detach() for k, v in loss_dict.items()}

    def training_step_latents(self, x, src_label, x0, lr, weight_decay=1e-2):
        loss, loss_dict = self(x, src_label, x0)
        self.trainer().adamw_step(lr, weight_decay=weight_decay)
        return loss, loss_dict
