"""CPU: the algebra behind csrc/winograd.hip, checked against torch's own conv2d in float64 -- the weight packers' constants
(ops._WINO_G, ops._UP_TAPS) and the transform matrices written in the kernels' header comment."""
import numpy as np
import torch
import torch.nn.functional as F

from conftest import rnd


def test_winograd_f2x2_3x3_identity():
    """Y = A^T [ (G g G^T) . (B^T d B) ] A summed over input channels equals the pad-1 cross-correlation nn.Conv2d computes."""
    from dsml_thesis_amd import ops
    G = torch.tensor(ops._WINO_G, dtype=torch.float64)
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    cin, cout, h, w = 5, 3, 6, 8
    x, g = rnd(1, 1, cin, h, w).double(), rnd(2, cout, cin, 3, 3).double()
    ref = F.conv2d(x, g, padding=1)[0]
    xp = F.pad(x, (1, 1, 1, 1))[0]
    U = torch.einsum("ia,ocab,jb->ijco", G, g, G)                        # what ops.pack_winograd stores as [16][cin][cout]
    out = torch.zeros(cout, h, w, dtype=torch.float64)
    for ty in range(h // 2):
        for tx in range(w // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]              # patch with its top-left at (2ty-1, 2tx-1)
            V = torch.einsum("ia,cab,jb->ijc", BT, d, BT)
            M = torch.einsum("ijc,ijco->ijo", V, U)
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = torch.einsum("ia,abo,jb->oij", AT, M, AT)
    torch.testing.assert_close(out, ref, rtol=1e-12, atol=1e-12)


def test_upsample_conv_is_four_2x2_tap_phase_convolutions():
    """nearest-x2 upsampling + Conv2d 3x3 (pad 1) == for each output parity (a, b) a 2x2-tap convolution of the low-resolution
    input at offsets (a-1+i, b-1+j) whose taps carry the summed 3x3 weights (ops._UP_TAPS)."""
    from dsml_thesis_amd import ops
    cin, cout, h, w = 4, 3, 5, 6
    x, g = rnd(3, 1, cin, h, w).double(), rnd(4, cout, cin, 3, 3).double()
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), g, padding=1)[0]
    xp = F.pad(x, (1, 1, 1, 1))[0]                                        # low-resolution zero padding: index + 1
    out = torch.zeros(cout, 2 * h, 2 * w, dtype=torch.float64)
    for a in range(2):
        for b in range(2):
            for i in range(2):
                for j in range(2):
                    wsum = sum(g[:, :, dy, dx] for dy in ops._UP_TAPS[a][i] for dx in ops._UP_TAPS[b][j])    # [cout][cin]
                    src = xp[:, a + i:a + i + h, b + j:b + j + w]         # x[y + a - 1 + i][x + b - 1 + j]
                    out[:, a::2, b::2] += torch.einsum("oc,chw->ohw", wsum, src)
    torch.testing.assert_close(out, ref, rtol=1e-12, atol=1e-12)


def test_folded_layernorm_identity():
    """LN(x) W + b == rstd (x W' - mean colsum(W')) + (beta^T W + b) with W' = diag(gamma) W (LDMK_TF_LAYERNORM_FOLDED)."""
    K, N, M = 24, 7, 11
    x, W_, b = rnd(5, M, K).double() + 0.7, rnd(6, K, N).double(), rnd(7, N).double()
    gamma, beta = 1 + 0.2 * rnd(8, K).double(), 0.3 * rnd(9, K).double()
    ref = F.layer_norm(x, (K,), gamma, beta, 1e-5) @ W_ + b
    mean = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    Wp = gamma[:, None] * W_
    out = rstd * (x @ Wp - mean * Wp.sum(0)) + (beta @ W_ + b)
    torch.testing.assert_close(out, ref, rtol=1e-11, atol=1e-11)


def test_gelu_erfc_form_matches_exact_gelu():
    """csrc/ldmk_common.h gelu_erf_f restated in float32 numpy: max abs error against float64 erf GELU below 5e-7."""
    from scipy.special import erf
    f = np.float32
    g = np.linspace(-12, 12, 400001).astype(f)
    z = np.abs(g) * f(0.70710678)
    t = (f(1) / (f(0.3275911) * z + f(1))).astype(f)
    p = (t * f(0.5 * 1.061405429) + f(0.5 * -1.453152027)).astype(f)
    for c in (0.5 * 1.421413741, 0.5 * -0.284496736, 0.5 * 0.254829592):
        p = (p * t + f(c)).astype(f)
    p = (p * t).astype(f)
    h = (g * (p * np.exp2((g * g * f(-0.72134752)).astype(f)).astype(f)).astype(f)).astype(f)
    y = np.maximum(g, f(0)) - np.abs(h)
    ref = 0.5 * g.astype(np.float64) * (1 + erf(g.astype(np.float64) / np.sqrt(2)))
    assert np.abs(y - ref).max() < 5e-7
