"""Condition encoders used by the shipped configs, with the reference's parameter names.  The class embedders are
table lookups (nn.Embedding); the audio window encoder runs as one fused HIP kernel (ldmk_audio_attention):
  ClassEmbedder3            face_reenactment/ldm/modules/encoders/modules.py:68-94
  ClassEmbedder             talking_face/ldm/modules/encoders/modules.py:44-73
  Conv1DTemporalAttention   talking_face/ldm/modules/encoders/modules.py:76-113
"""
import torch
import torch.nn as nn


class ClassEmbedder3(nn.Module):
    def __init__(self, embed_dim, n_classes, key="class_label", p_uncond=0.2):
        super().__init__()
        self.p_uncond, self.key = p_uncond, key
        self.embedding = nn.Embedding(n_classes, embed_dim)
        self.uncond_embedding = nn.Embedding(1, embed_dim)

    def forward(self, batch, training=False, key=None):
        key = self.key if key is None else key
        c = batch[key][:, None]
        if training and torch.rand(1) < self.p_uncond:
            return self.uncond_embedding(torch.zeros_like(c))
        return self.embedding(c)


class ClassEmbedder(nn.Module):
    def __init__(self, embed_dim, n_classes, key="class_label", p_uncond=0.2):
        super().__init__()
        self.p_uncond, self.n_classes, self.key = p_uncond, n_classes, key
        self.embedding = nn.Embedding(n_classes + 1, embed_dim)      # last row = trainable null class

    def forward(self, batch, training=False, key=None):
        key = self.key if key is None else key
        c = batch[key][:, None]
        if training and torch.rand(1) < self.p_uncond:
            c = torch.full_like(c, self.n_classes)
        return self.embedding(c)


class Conv1DTemporalAttention(nn.Module):
    """(b, T, 768) window of wav2vec2 features -> attention-pooled (b, 1, 768)."""

    def __init__(self, seq_len, subspace_dim=768, subspace2hidden=False, hidden_dim=None):
        super().__init__()
        if subspace2hidden:
            raise NotImplementedError("Conv1DTemporalAttention: subspace2hidden is off in the shipped config")
        self.seq_len, self.subspace_dim = seq_len, subspace_dim
        chans = [subspace_dim, 192, 64, 16, 4, 1]
        layers = []
        for i in range(5):
            layers += [nn.Conv1d(chans[i], chans[i + 1], kernel_size=3, stride=1, padding=1, bias=True),
                       nn.LeakyReLU(0.02, True)]
        self.attentionConvNet = nn.Sequential(*layers)
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def _pack(self):
        """Conv1d weights [cout][cin][3] -> [3][cin][cout] (coalesced per output channel) + device pointer tables."""
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if getattr(self, "_sig", None) == sig:
            return
        convs = [self.attentionConvNet[i] for i in range(0, 10, 2)]
        self._w = [c.weight.detach().float().permute(2, 1, 0).contiguous() for c in convs]
        self._b = [c.bias.detach().float().contiguous() for c in convs]
        dev = self._w[0].device
        self._wp = torch.tensor([t.data_ptr() for t in self._w], dtype=torch.int64, device=dev)
        self._bp = torch.tensor([t.data_ptr() for t in self._b], dtype=torch.int64, device=dev)
        self._lw = self.attentionNet[0].weight.detach().float().contiguous()
        self._lb = self.attentionNet[0].bias.detach().float().contiguous()
        self._sig = sig

    @torch.no_grad()
    def forward(self, x):
        """x: (b, T, 768) CUDA float32 -> (b, 1, 768)."""
        from . import lib as L
        if not x.is_cuda:
            raise L.LdmkError("Conv1DTemporalAttention: CUDA tensors only (no CPU fallback)")
        b, T, dim = x.shape
        assert T == self.seq_len and dim == self.subspace_dim
        self._pack()
        xc = x.float().contiguous()
        out = torch.empty(b, dim, device=x.device, dtype=torch.float32)
        L.call("ldmk_audio_attention", xc.data_ptr(), b, T, dim, self._wp.data_ptr(), self._bp.data_ptr(),
               self._lw.data_ptr(), self._lb.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out.unsqueeze(1)


def audio_attention_backward(module, x, dout):
    """Parameter gradients of `Conv1DTemporalAttention` (x: (b,T,768) window, dout: (b,1,768) gradient of its output)
    on the HIP kernel `ldmk_audio_attention_bwd`; written to the `.grad` of the module's parameters (torch layouts)."""
    from . import lib as L
    from . import train_ops as T_
    b, T, dim = x.shape
    module._pack()
    per = L.load().ldmk_audio_attention_grad_elems(T, dim)
    gs = torch.empty(b, per, device=x.device, dtype=torch.float32)
    L.call("ldmk_audio_attention_bwd", x.float().contiguous().data_ptr(), dout.float().contiguous().data_ptr(), b, T, dim,
           module._wp.data_ptr(), module._bp.data_ptr(), module._lw.data_ptr(), module._lb.data_ptr(), gs.data_ptr(),
           torch.cuda.current_stream().cuda_stream)
    g = T_.colsum(gs).view(-1)                     # sum over the batch
    chans = [dim, 192, 64, 16, 4, 1]
    convs = [module.attentionConvNet[i] for i in range(0, 10, 2)]
    off = 0
    for l, c in enumerate(convs):
        n_ = 3 * chans[l] * chans[l + 1]
        c.weight.grad = g[off:off + n_].view(3, chans[l], chans[l + 1]).permute(2, 1, 0).contiguous()
        off += n_
    for l, c in enumerate(convs):
        c.bias.grad = g[off:off + chans[l + 1]].clone()
        off += chans[l + 1]
    lin = module.attentionNet[0]
    lin.weight.grad = g[off:off + T * T].view(T, T).clone()
    lin.bias.grad = g[off + T * T:off + T * T + T].clone()


def mask_lower_face_(images, first_masked_row, value=-1.0):
    """In place: images[n, :, y >= first_masked_row[n], :] = value (MEADBase3 sampling mask, custom.py:375-389;
    first_masked_row = int(min(mouth landmark y)) - 5).  images: (n, c, h, w) CUDA float32."""
    from . import lib as L
    n, c, h, w = images.shape
    y0 = torch.as_tensor(first_masked_row, dtype=torch.int32, device=images.device).contiguous()
    L.call("ldmk_mask_rows", images.data_ptr(), y0.data_ptr(), n, c, h, w, float(value),
           torch.cuda.current_stream().cuda_stream)
    return images
