"""Condition encoders used by the shipped configs.  They run once per sample()/frame on a handful of rows
(negligible next to 200 UNet evaluations), so they stay plain PyTorch-ROCm modules with the
reference's parameter names:
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

    @staticmethod
    def _conv1d_k3(x, conv):
        """Conv1d(k=3, pad=1) as unfold + one GEMM: bitwise run-to-run reproducible (MIOpen's conv1d picks its
        algorithm by a first-call search and differs in the last bit between calls), same parameters/keys."""
        b, cin, T = x.shape
        xp = torch.nn.functional.pad(x, (1, 1))
        cols = torch.stack([xp[:, :, k:k + T] for k in range(3)], dim=-1)          # (b, cin, T, 3)
        cols = cols.permute(0, 2, 1, 3).reshape(b * T, cin * 3)
        out = torch.nn.functional.linear(cols, conv.weight.reshape(conv.out_channels, cin * 3), conv.bias)
        return out.view(b, T, conv.out_channels).transpose(1, 2)

    def forward(self, x):
        b = x.shape[0]
        xt = torch.transpose(x, 1, 2)
        h = xt
        for i in range(0, 10, 2):
            h = torch.nn.functional.leaky_relu(self._conv1d_k3(h, self.attentionConvNet[i]), 0.02)
        att = self.attentionNet(h.reshape(b, self.seq_len)).view(b, self.seq_len, 1)
        return torch.bmm(xt, att).view(b, self.subspace_dim).unsqueeze(1)
